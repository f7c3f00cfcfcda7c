"""Reference import path ``validate`` -> egopack_amd.validate (same function names and signatures)."""
from egopack_amd.validate import validate, validate_lta, validate_pnr  # noqa: F401
