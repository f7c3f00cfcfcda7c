"""Two rank processes stepping together (sharing the box's one GPU, gloo transport) against one process on the global
batch: tools/two_rank_check.py.  Covers the N > 1 step itself -- sharded batches, GradSync(world 2), FlatAdam's 1 / world
scale -- and the exact cross-rank graph-LayerNorm statistics (SURVEY 8e caveat 1)."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parents[1]


def test_two_ranks_with_exact_graph_ln_equal_one_process_on_the_global_batch():
    r = subprocess.run([sys.executable, str(REPO / "tools" / "two_rank_check.py")], capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, f"no verdict line (rc {r.returncode}):\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    out = json.loads(lines[-1])
    e, l = out["exact"], out["local"]
    # exact mode: the two-rank run IS the one-process run on the global batch, up to f32 summation order
    assert e["objective_rel"] <= 1e-6, out
    assert e["grad_rel"] <= 2e-3, out  # (a few ReLU / LeakyReLU gates at rounding distance from zero: see test_gpu_configs.py)
    assert e["param_frac_within_2e-4"] >= 0.999, out
    # default mode (per-rank statistics) is a different computation: the mode, not luck, makes the runs agree
    assert l["grad_rel"] >= 100 * e["grad_rel"] and l["param_frac_within_2e-4"] < 0.99, out
    assert e["ranks_bit_identical"] and l["ranks_bit_identical"], out
    # benchmark mode: both ranks capture the step as staged hipGraphs; replayed steps = eagerly issued steps, bit for bit
    rp = out["bf16_replay"]
    assert rp["capture"] == "staged graphs" and rp["ranks_bit_identical"] and rp["replayed_vs_eager_max_abs"] == 0.0, out
    assert r.returncode == 0, out


def test_bench_runs_two_ranks_through_the_staged_graphs():
    """``bench.py --gpus 2`` end to end with two REAL ranks (both on the box's one GPU, gloo transport, --one-gpu-gloo):
    the ranks are spawned by bench.py itself, the step is captured as staged graphs on both (no silent fallback), the
    timing protocol and the per-kernel leg complete on every rank (a rank 0 taking its eager profiling steps alone used to
    wait for its peers for ever), and rank 0's JSON line is the last line of stdout."""
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--one-gpu-gloo", "--steps", "4", "--warmup", "2",
                        "--strict-capture"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    last = r.stdout.strip().splitlines()[-1]
    out = json.loads(last)
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2"
    assert out["config"]["capture"] == "staged graphs" and out["config"]["capture_fallbacks"] == []
    assert out["config"]["grad_allreduce"] == "f32" and "ONE GPU" in out["config"]["transport"]
    assert out["roofline"] and "error" not in out["roofline"], out["roofline"]
    assert out["config"]["global_batch"] == 2 * 3 * 64
