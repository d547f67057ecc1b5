"""CPU: the reference's YAML `class_path`/`init_args` trees build through refign_amd.config without modification (the
model section of configs/cityscapes_darkzurich/refign_hrda_star.yaml is embedded verbatim in bench.REF_CFG), with the
reference's trainable-parameter counts (85.69 M HRDA / 85.16 M DAFormer) and optimiser groups."""
import copy

import torch


def _build(use_hrda):
    import bench
    from refign_amd import config
    cfg = copy.deepcopy(bench.REF_CFG)
    cfg["model"]["init_args"]["use_hrda"] = use_hrda
    over = {"backbone.init_args.pretrained": None, "alignment_backbone.init_args.pretrained": None,
            "alignment_head.init_args.pretrained": None}
    return config.build_model(cfg, over)


def test_hrda_config_builds_with_reference_param_count():
    m = _build(True)
    n = sum(p.numel() for p in m.parameters() if p.requires_grad)
    assert abs(n / 1e6 - 85.686) < 0.01
    groups = m.optimizer_parameters()
    assert [g["name"] for g in groups] == ["head_weight", "head_bias", "backbone_weight", "backbone_bias"]
    assert groups[2]["lr"] == 0.1 * groups[0]["lr"] and groups[1]["weight_decay"] == 0
    assert sum(p.numel() for g in groups for p in g["params"]) == n
    # frozen: alignment nets, EMA teacher, ImageNet encoder
    assert not any(p.requires_grad for p in m.alignment_head.parameters())
    assert not any(p.requires_grad for p in m.m_backbone.parameters())
    assert not any(p.requires_grad for p in m.imnet_backbone.parameters())
    (opt,), (sch,) = m.configure_optimizers()
    assert isinstance(opt, torch.optim.AdamW) and type(sch).__name__ == "LinearWarmupPolynomialLR"


def test_daformer_config_param_count():
    m = _build(False)
    assert abs(sum(p.numel() for p in m.parameters() if p.requires_grad) / 1e6 - 85.155) < 0.01
    assert m.hrda_scale_attention is None


def test_train_mode_keeps_frozen_nets_in_eval():
    m = _build(False).train()
    assert m.backbone.training and m.m_backbone.training          # teacher in train mode: reference quirk D7/D9
    assert not m.alignment_backbone.training and not m.alignment_head.training and not m.imnet_backbone.training


# ---- the reference's own YAML files, where they exist (authoring container only: /root/reference is absent on the GPU box)
import glob  # noqa: E402
import os  # noqa: E402

import pytest  # noqa: E402

_REF_CONFIGS = sorted(glob.glob("/root/reference/configs/**/*.yaml", recursive=True))


@pytest.mark.skipif(not _REF_CONFIGS, reason="reference checkout not present")
@pytest.mark.parametrize("path", _REF_CONFIGS, ids=[os.path.relpath(p, "/root/reference/configs") for p in _REF_CONFIGS])
def test_reference_yaml_builds_unmodified(path):
    """Every configs/**/*.yaml of the reference goes through config.load_config + build_model untouched (only the
    pretrained-checkpoint paths are nulled: the files cannot be downloaded here).  DeepLabV2 / ResNet configs are outside
    the hot path (out of scope) and must fail with a clear message, not a ModuleNotFoundError."""
    from refign_amd import config
    cfg = config.load_config(path)
    assert {"model"} <= set(cfg)
    init = cfg["model"]["init_args"]
    over = {}
    for k in ("backbone", "alignment_backbone", "alignment_head", "head"):
        if isinstance(init.get(k), dict) and "pretrained" in init[k].get("init_args", {}):
            over[f"{k}.init_args.pretrained"] = None
    if "pretrained" in init:
        over["pretrained"] = None
    if "deeplabv2" in path:
        with pytest.raises(config.OutOfScopeError, match="out of scope"):
            config.build_model(cfg, over)
        return
    m = config.build_model(cfg, over)
    want = {"models.DomainAdaptationSegmentationModel": "DomainAdaptationSegmentationModel",
            "models.AlignmentModel": "AlignmentModel"}[cfg["model"]["class_path"]]
    assert type(m).__name__ == want
    if want == "DomainAdaptationSegmentationModel":
        assert m.use_refign == bool(init.get("use_refign", False)) if hasattr(m, "use_refign") else True
        (opt,), (sch,) = m.configure_optimizers()
        assert type(opt).__name__ == cfg["optimizer"]["class_path"].rsplit(".", 1)[-1]


def test_bench_stall_guard_exits_and_says_where(tmp_path):
    """bench.py, N > 1: a run that stops making progress exits non-zero (17) and says where it stopped instead of hanging the
    node.  (No second attempt in another configuration: the multi-rank default is the one that needs none, bn.ddp_mode.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "p = [time.monotonic() - 100.0, 'warm-up step 1']\n"
            "bench._stall_guard(1, 2, p)\n"
            "time.sleep(30)\n" % root)
    env = dict(os.environ, RFN_BENCH_STALL_S="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 17, r.stderr
    assert "no progress" in r.stderr and "warm-up step 1" in r.stderr and "rank 1/2" in r.stderr


def test_trainer_stall_guard_watches_the_step():
    """The same guard inside the trainer (not only in bench.py): refign_amd.trainer.StallGuard, started by Trainer for a
    world of more than one rank, with Trainer.step as its heartbeat -- here driven directly: a heartbeat that stops ends the process
    with code 17 and names the last phase."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r)\n"
            "from refign_amd.trainer import StallGuard\n"
            "g = StallGuard(3, 8)\n"
            "g.note('step 41 queued')\n"
            "time.sleep(30)\n" % root)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RFN_STALL_S="1"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 17, r.stderr
    assert "rank 3/8" in r.stderr and "step 41 queued" in r.stderr


def test_ddp_mode_selector(monkeypatch):
    """RFN_DDP_MODE: the N > 1 default is `torch` (every exchange through torch.distributed); the direct-RCCL modes are opt-in;
    anything else is refused."""
    import pytest
    from refign_amd import bn, rccl
    monkeypatch.delenv("RFN_DDP_MODE", raising=False)
    assert bn.ddp_mode() == "torch" and not rccl.enabled()
    for m in ("direct", "direct3"):
        monkeypatch.setenv("RFN_DDP_MODE", m)
        assert bn.ddp_mode() == m and rccl.enabled()
    monkeypatch.setenv("RFN_DDP_MODE", "fastest")
    with pytest.raises(RuntimeError, match="RFN_DDP_MODE"):
        bn.ddp_mode()


def test_correlation_channel_split_heuristic(monkeypatch):
    """correlation._channel_splits (host logic of the tiny-map path, csrc/corr.hip launch_corr9_split): the split applies
    to maps of a few 8x64 tiles only, chunks are multiples of 8 channels and at least 16, powers of two up to 8; the
    bench's levels 1 and 2 keep the one-kernel path; the switches."""
    from refign_amd.correlation import _channel_splits as f
    monkeypatch.delenv("RFN_CORR_SPLIT", raising=False)
    monkeypatch.delenv("RFN_CORR_SPLITS", raising=False)
    assert f(2, 256, 32, 32) == 4                      # K4 level 3: chunks of 64 channels, joined inside the launch (round 5)
    assert f(2, 512, 32, 32) == 8 and f(2, 256, 64, 64) == 2 and f(2, 48, 16, 32) == 2   # (48: two-launch form, chunks of 24)
    assert f(2, 128, 270, 480) == 1 and f(2, 256, 135, 240) == 1
    assert f(2, 256, 32, 30) == 1                      # W % 4
    assert f(1, 20, 16, 16) == 1                       # C % 8
    for (b, c, h, w) in [(1, 128, 9, 12), (2, 64, 17, 64), (1, 512, 16, 16), (3, 32, 8, 4), (1, 16, 8, 8), (4, 24, 8, 64)]:
        s = f(b, c, h, w)
        assert s in (1, 2, 4, 8) and c % s == 0 and (c // s) % 8 == 0 and (s == 1 or c // s >= 16), (b, c, h, w, s)
    monkeypatch.setenv("RFN_CORR_SPLIT", "0")
    assert f(2, 256, 32, 32) == 1
    monkeypatch.setenv("RFN_CORR_SPLIT", "1")
    monkeypatch.setenv("RFN_CORR_SPLITS", "4")
    assert f(2, 256, 135, 240) == 4 and f(2, 24, 32, 32) == 1     # forced, unless the chunks would not be multiples of 8
