"""End-to-end run of the two entry points on the GPU with tiny synthetic datasets: multi-task pre-training
writes a checkpoint with the reference's key layout, the EgoPack phase resumes from it, builds the
prototype banks and trains the novel task."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(600)
def test_main_temporal_then_main_egopack(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import main_egopack
    import main_temporal
    common = ["k=1", "batch_size=4", "num_epochs=2", "synthetic_samples=16", "model.hidden_size=64",
              "model.temporal_pooling.hidden_size=64", "oscc_feat_size=64", f"checkpoint_dir={tmp_path}", "save_model=True",
              "compute=f32", "optimizer.lr=1e-3"]
    main_temporal.main(common + ["enabled_tasks=[ar,lta,pnr]"])
    ckpt_path = tmp_path / "MTL_ar-lta-pnr" / "checkpoint.pth"
    ckpt = torch.load(ckpt_path, weights_only=False)
    assert {"temporal_graph", "task/recognition", "task/oscc", "task/lta", "task/pnr", "epoch"} <= set(ckpt)
    assert "net.module_0.lin_l.weight" in ckpt["temporal_graph"] and ckpt["epoch"] == 2
    assert all(torch.isfinite(v).all() for v in ckpt["temporal_graph"].values())
    common[3] = "synthetic_samples=256"  # build_graphone reads the AR split with batch 256, drop_last=True
    main_egopack.main(common + ["enabled_tasks=[oscc]", "enable_graphone=True", f"resume_from={ckpt_path}", "graphone.k=4",
                                "graphone.depth=2", "graphone.residual=True", "graphone.hidden_size=64",
                                "+graphone.features_size=64", "artifact_prefix=EGO"])
    ego = torch.load(tmp_path / "EGO_egopack_oscc" / "checkpoint.pth", weights_only=False)
    assert "graphone" in ego and any(k.startswith("embeddings.ar") for k in ego["graphone"])
    assert any(k.startswith("conv_stages.pnr.1.module_3") for k in ego["graphone"])
    # the backbone moved (backprop_temporal_graph defaults to true), the frozen banks did not
    moved = sum((ego["temporal_graph"][k] - ckpt["temporal_graph"][k]).abs().sum() for k in ckpt["temporal_graph"])
    assert moved > 0
