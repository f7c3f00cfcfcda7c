from egopack_amd.criterion import BCEWithLogitsNone, CrossEntropyNone, MetricSelectorWrapper  # noqa: F401
