"""CPU oracle for the EgoPack training hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (CPU, fp32/fp64) restatement of the arithmetic on the
reference's hot path.  It exists to *check* the HIP path:

  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
    import it; nothing under ``egopack_amd/`` or the top-level ``models/`` mirror does;
  * the product path never falls back to it -- without the HIP library the product raises.

Pinning status (see DESIGN.md "Oracle"):
  * Everything the reference itself owns (TRNPooling, task heads, losses, cos_dissimilarity,
    Graph.forward / GraphONE control flow, build_graphone, LTATemporalConnectivity,
    multiloader) is PINNED: ``oracle/make_golden.py`` imports the reference's own modules
    from /root/reference in the build container and stores their outputs in tests/golden/.
  * The leaf ops of the absent third-party dependency torch_geometric==2.3.0
    (+ torch_cluster 1.6.1 / torch_scatter 2.1.1; reference environment.yml:174,184,188) --
    SAGEConv, graph-mode LayerNorm, PositionalEncoding, global_max_pool, scatter,
    add_remaining_self_loops, radius_graph, coalesce, Batch collation -- are restated from
    that release's published semantics in ``oracle/pyg_ops.py``.  The reference holds no
    tests or vectors for them, so for those leaf ops: **parity unpinned** beyond the
    hand-derived known-answer tests in tests/test_oracle_known_answers.py.
  * Validation meters (``oracle/meters.py``): the reference's own numpy top-k accuracy / recall,
    its PNR localisation and LTA edit-distance bookkeeping and its validate.py loops are PINNED by
    tests/golden/meters.pt and tests/golden/validate.pt (``oracle/make_golden_meters.py``); the absent packages
    torchmetrics 0.11 / editdistance 0.6 are restated from their published definitions: **parity unpinned** for
    those leaves beyond the known-answer tests in tests/test_meters_cpu.py.
  * Input pipeline: the reference's sampling functions and its four datasets' ``get`` methods are
    PINNED by tests/golden/sampling.pt and tests/golden/pipeline.pt (``oracle/make_golden_sampling.py``,
    ``oracle/make_golden_pipeline.py``).
"""
