"""Host-side mirror of the reference interface: registry, config, misc helpers."""
import os

import pytest
import torch

from utils import registry
from utils.config import EasyDict, builtin_cfg
from utils import misc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_registry_contract():
    R = registry.Registry('things')

    @R.register_module()
    class Foo:
        def __init__(self, cfg):
            self.v = cfg.v

    assert R.get('Foo') is Foo and 'Foo' in R and len(R) == 1
    assert R.build(EasyDict(NAME='Foo', v=3)).v == 3
    with pytest.raises(KeyError, match='already has a class'):
        R.register_module()(Foo)
    with pytest.raises(KeyError, match='things registry holds no class called Bar'):
        R.build(EasyDict(NAME='Bar'))
    with pytest.raises(KeyError, match='NAME'):
        R.build(EasyDict(v=1))
    with pytest.raises(TypeError):
        R.build([1, 2])
    with pytest.raises(AttributeError, match='Foo'):       # constructor errors are re-raised naming the class
        R.build(EasyDict(NAME='Foo'))


def test_builtin_config_fields():
    cfg = builtin_cfg('unify_modelnet_cls')
    m = cfg.model
    assert m.NAME == 'Point_MAE_unify' and m.num_group == 64 and m.group_size == 32
    assert m.transformer_config.trans_dim == 384 and m.transformer_config.depth == 12
    assert m.prompter_config['rectify_depth'] == 3 and dict(**m.prompter_config)['downstream_prompts_num'] == 10
    assert m.gather_idx is False and m.prompt_propagation_after is True


def test_misc_helpers():
    assert misc.peft_detect('blocks.blocks.0.bnorm.weight', ['bnorm']) and not misc.peft_detect('norm.weight', ['bnorm'])
    p = torch.rand(2, 100, 3)
    n = misc.lidar_noise(p, 48, low=1.2, scale=1.5)
    assert n.shape == (2, 48, 3)
    g = misc.gaussian_noise([2, 24, 3], scale=0.1)
    r = g.norm(dim=-1)
    assert g.shape == (2, 24, 3) and (r > 0.5).all() and (r < 1.5).all()


def test_batch_counters_count_every_call_of_a_shared_module():
    """The per-forward counter list may hold one BatchNorm counter several times (shared patch embedding)."""
    import torch
    from models import upp_layers as L
    a, b = torch.zeros((), dtype=torch.long), torch.zeros((), dtype=torch.long)
    L.begin_forward(torch.device('cpu'), True)
    for t in (a, b, a, a):
        L.bump_counter(t)
    L.end_forward()
    assert int(a) == 3 and int(b) == 1
    L.bump_counter(b)                      # outside a forward: immediate
    assert int(b) == 2


def test_bench_gpus_flag_launches_that_many_ranks(tmp_path):
    """`python bench.py --gpus 2` alone must start 2 ranks (torch.distributed.run child, 127.0.0.1 rendezvous), rank 0 prints ONE
    JSON line with n_gpus == 2, and a WORLD_SIZE that contradicts --gpus is refused.  --selftest-launch keeps the GPU out of it."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, UPP_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--selftest-launch"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and abs(rec["max_over_ranks_s"] - 0.002) < 1e-12
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-launch"],
                         env=dict(env, WORLD_SIZE="1"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in bad.stderr


def test_transposed_weight_cache_tracks_versions_views_and_refresh():
    """functional.TRANSPOSED (W^T copies for the data-gradient GEMMs of frozen layers): one copy per weight or column window,
    refreshed IN PLACE when the owner's version counter moves (graphs keep the address), refresh() re-copies everything."""
    from upp_hip.functional import _TransposedWeights
    c = _TransposedWeights()
    w = torch.arange(12.0).view(3, 4)
    t1 = c.get(w)
    assert torch.equal(t1, w.t()) and t1.is_contiguous()
    assert c.get(w) is t1                                  # cached
    w.mul_(2.0)                                            # in-place update (load_state_dict): version moves
    t2 = c.get(w)
    assert t2 is t1 and torch.equal(t1, w.t())             # same storage, new contents
    big = torch.arange(24.0).view(3, 8)
    win = big[:, 4:]                                       # a column window of a parameter (Conv1d(512,512) split in the encoder)
    tw = c.get(win)
    assert torch.equal(tw, win.t()) and c.get(big[:, 4:]) is tw and c.get(big[:, :4]) is not tw
    big.add_(1.0)
    assert torch.equal(c.get(big[:, 4:]), big[:, 4:].t())
    with torch.no_grad():
        w.copy_(torch.ones(3, 4))
    c.entries[next(iter(c.entries))][1] = w._version       # pretend nobody noticed: refresh() must still re-copy
    c.refresh()
    assert torch.equal(c.get(w), torch.ones(4, 3))


def test_padded_first_conv_weight_is_refreshed_in_place():
    """Encoder._w1_k32: the zero-padded K = 32 image of the frozen Conv1d(3,128) weight keeps its storage when the weight changes (a
    captured step and the W^T copy of the data-gradient GEMM hold its address); functional.refresh_caches(model) re-copies it and then
    the transposes, in that order, even when nobody noticed the change."""
    from models.upp_layers import Encoder
    from upp_hip import functional as HF
    enc = Encoder(384)
    for p in enc.parameters():
        p.requires_grad_(False)
    w = enc.first_conv[0].weight
    first = enc._w1_k32()
    ptr = first.data_ptr()
    assert first.shape == (128, 32) and torch.equal(first[:, :3], w.squeeze(-1)) and float(first[:, 3:].abs().sum()) == 0.0
    wt = HF.TRANSPOSED.get(first)
    with torch.no_grad():
        w.mul_(2.0)                                        # load_state_dict: same storage, new contents
    again = enc._w1_k32()
    assert again.data_ptr() == ptr and torch.equal(again[:, :3], w.squeeze(-1))
    with torch.no_grad():
        w.add_(1.0)
    enc._w1p_key = (w.data_ptr(), w._version)              # pretend nobody noticed (the captured-step case)
    HF.refresh_caches(enc)
    assert enc._w1p.data_ptr() == ptr and torch.equal(enc._w1p[:, :3], w.squeeze(-1))
    assert HF.TRANSPOSED.get(enc._w1p) is wt and torch.equal(wt, enc._w1p.t())


def test_declined_fused_paths_are_reported_once(monkeypatch, capsys):
    from upp_hip import functional as HF
    monkeypatch.setenv("UPP_VERBOSE", "1")
    HF._declined.discard(("unit-test site", "reason"))
    HF.note_declined("unit-test site", "reason")
    HF.note_declined("unit-test site", "reason")
    err = capsys.readouterr().err
    assert err.count("unit-test site") == 1 and "fused path declined" in err


def test_capture_state_snapshot_restores_in_place():
    """train._TrainingState: what the graph-capture warm-up uses to leave model and optimizer untouched."""
    from upp_hip.train import _TrainingState
    m = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3))
    opt = torch.optim.AdamW(m.parameters(), lr=0.1)
    flat = torch.zeros(5)
    m(torch.randn(8, 4)).sum().backward()
    opt.step()
    keep = _TrainingState(m, opt, flat)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    ptrs = [p.data_ptr() for p in m.parameters()]
    step_before = opt.state_dict()['state'][0]['step'].clone()
    for _ in range(2):                                     # "warm-up": trains, moves BatchNorm statistics and counters
        opt.zero_grad()
        m(torch.randn(8, 4)).sum().backward()
        opt.step()
        flat += 1
    keep.restore()
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k]), k
    assert [p.data_ptr() for p in m.parameters()] == ptrs and float(flat.abs().sum()) == 0.0
    assert torch.equal(opt.state_dict()['state'][0]['step'], step_before)


def test_committed_pmc_counters_were_taken_on_the_kernels_this_build_launches():
    """bench.py attaches the committed rocprofv3 --pmc counters of a labelled Linear launch only if they were measured on the kernel the
    present build launches for that label (name with ALL template arguments).  Round 4's driver line carried traffic = null for every
    label because a sixth template argument had been added after the counters were taken; this is the commit-time guard: the tile choice
    is a host function of the library, so the join can be checked without a GPU."""
    import importlib.util
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    raw = bench._pmc_raw()
    stale, missing = [], []
    for label, M, N, K, epi in bench.STEP_LINEAR_SHAPES:
        kname = bench.sb_kernel_name(M, N, K, epi)
        assert kname.startswith("linear_sb_kernel<") and kname.count(",") == 5, kname      # six template arguments, as rocprofv3 prints them
        if "linear:" + label not in raw:
            missing.append(label)
        elif bench.pmc_linear_entry(raw, label, kname) is None:
            stale.append("%s: counters on %s, build launches %s" % (label, raw["linear:" + label].get("kernel"), kname))
    assert not stale, stale
    assert not [m for m in missing if "@" not in m], missing          # the eight labels of one block at M = 2400: always present
    # every labelled shape of the step resolves back to its (first) label
    for label, M, N, K, epi in bench.STEP_LINEAR_SHAPES:
        assert bench.linear_label(M, N, K, epi) in [l for l, m, n, k, e in bench.STEP_LINEAR_SHAPES if (m, n, k, e) == (M, N, K, epi)]


def test_torch_kernels_the_steps_launch_carry_no_affected_packed_f32_routing():
    """Round 4 found v_pk_{add,mul,fma}_f32 with op_sel routing src1's HIGH half to the low lane returning wrong low halves beside a
    bf16-MFMA workgroup; libupp_hip.so is built without packed f32 (tests/test_abi.py), torch's own element-wise kernels -- which the
    pipelined default still launches beside the split-bf16 GEMMs -- are not ours to rebuild.  tools/torch_pk_scan.py disassembled every
    gfx950 code object of this image's libtorch_hip.so (profiles/r05_torch_pk_scan.txt; several minutes, so the scan is committed and this
    test reads it): the affected routing occurs ONLY in complex<float>, BFloat16-reduction and geometric-distribution instantiations --
    none of the f32 add / mul / fill / cat / copy / layer-norm / reduce / uniform kernels a step launches (profiles/r0N_glue_census*.txt)."""
    from conftest import ROOT
    text = open(os.path.join(ROOT, "profiles", "r05_torch_pk_scan.txt")).read()
    head, _, listed = text.partition("instantiations with the affected routing:")
    assert "gfx950 code objects" in head and "v_pk_{add,mul,fma}_f32" in head
    rows = [l for l in listed.splitlines() if l.strip()]
    assert rows, "the scan lists the affected instantiations it found"
    for l in rows:
        assert ("complex<float>" in l) or ("BFloat16" in l) or ("geometric_kernel" in l), l
    # per kernel family: the f32-only families the steps launch report 0 affected instructions
    for fam in ("FillFunctor", "CatArrayBatchedCopy", "direct_copy_kernel", "vectorized_layer_norm_kernel", "MaxOps<float>", "vectorized_gather_kernel",
                "BinaryFunctor<float, float, float, at::native::binary_internal::MulFunctor", "AUnaryFunctor<float, float, float, at::native::binary_internal::MulFunctor"):
        line = next(l for l in head.splitlines() if l.strip().startswith(fam[:60]))
        assert line.rstrip().endswith("affected routing: 0"), line


def test_configs0_cpu_plumbing_forward_runs_on_the_opt_in_torch_formulations(oracle_ops):
    """BASELINE configs[0]: Point_MAE_unify cls forward of one N = 1024 cloud on torch-CPU (utils.misc.fps + a torch.cdist kNN).  The
    product has no CPU path by default (ops raises); upp_hip.torch_cpu.enable() -- explicit, off by default -- serves CPU tensors with
    plain torch.  Here: (a) off -> the operators refuse CPU tensors; (b) on -> FPS picks and kNN lists equal the CPU oracle's on a
    seeded cloud and the model's logits equal the oracle-injected forward to 1e-5 (eval mode; the oracle is only the checker)."""
    import numpy as np
    import oracle
    import _seeded
    from models import build_model_from_cfg, upp_layers
    from upp_hip import functional as HF, torch_cpu
    import knn_cuda
    x = _seeded.noisy_clouds(1, 1024, seed=5)
    model = _seeded.fill(build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model)).eval()
    with torch.no_grad():
        want = model(x, completion_prompt=True, denoise=True, point_num=1024)      # (oracle_ops fixture: grouping on the C oracle)
    saved = dict(upp_layers.OPS)
    saved_fg = HF.fps_gather
    try:
        HF.fps_gather = _PRODUCT_FPS_GATHER                       # back to the product's own entry points (the fixture had replaced them)
        upp_layers.OPS["fps_gather"] = _PRODUCT_FPS_GATHER
        upp_layers.OPS["knn_group"] = _PRODUCT_KNN_GROUP
        assert not torch_cpu.enabled()
        with pytest.raises(RuntimeError):
            _PRODUCT_FPS_GATHER(x, 64)
        torch_cpu.enable()
        cen, idx = _PRODUCT_FPS_GATHER(x, 64)
        assert np.array_equal(idx.numpy(), oracle.fps(x.numpy(), 64))
        d, i = knn_cuda.KNN(32, transpose_mode=True)(x, cen)
        _, wi = oracle.knn(x.numpy(), cen.numpy(), 32, want_dist=False)
        assert np.array_equal(np.sort(i.numpy(), -1), np.sort(wi, -1))                 # neighbour SETS (in-list order is not pinned)
        wd, _ = oracle.knn(x.numpy(), cen.numpy(), 32, want_dist=True)                   # EUCLIDEAN distances, as KNN_CUDA / the HIP path
        np.testing.assert_allclose(np.sort(d.numpy(), -1), np.sort(wd, -1), rtol=2e-6, atol=1e-7)
        with torch.no_grad():
            got = model(x, completion_prompt=True, denoise=True, point_num=1024)
        np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-5, atol=2e-5)
    finally:
        torch_cpu.enable(False)
        upp_layers.OPS.clear()
        upp_layers.OPS.update(saved)
        HF.fps_gather = saved_fg


from upp_hip import functional as _HF_for_fallback_test          # noqa: E402  (the product's own entry points, captured before any fixture replaces them)
_PRODUCT_FPS_GATHER, _PRODUCT_KNN_GROUP = _HF_for_fallback_test.fps_gather, _HF_for_fallback_test.knn_group


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_bench_stdout_line_fits_the_drivers_record(tmp_path, monkeypatch, capsys):
    """Round 5's line was 22 KB and the driver's 8 KB tail lost metric / value / roofline (BENCH_r05.json parsed: null).  The line is
    now built by bench.compact_line from the full report: canned here from the committed round-5 report (the largest ever printed)
    with every list doubled, it must stay under 6,000 bytes, carry metric and value in its first 300 bytes, keep roofline + cpu_baseline,
    and the full report must land in the side file."""
    import json
    bench = _load_bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench.json")))
    full["roofline"]["by_shape"] = full["roofline"]["by_shape"] * 2
    full["kernels"].update({k + "_again": v for k, v in list(full["kernels"].items())})
    full.update(repeats=3, ms_per_step_min=4.2, ms_per_step_max=4.6)
    assert len(json.dumps(full)) > 30000
    compact, detail = bench.compact_line(full)
    text = json.dumps(compact)
    assert len(text) < 6000 and len(text) <= bench.LINE_BUDGET
    assert '"metric"' in text[:300] and '"value"' in text[:300]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_sequential", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "cpu_baseline", "rccl_ranks", "dist_backend", "repeats", "ms_per_step_min",
              "ms_per_step_max", "detail"):
        assert k in compact, k
    roof = compact["roofline"]
    for k in ("kernel", "bound", "ms", "launches", "achieved", "peak", "unit", "frac", "frac_of_f32_mfma_peak", "algorithmic_flops",
              "algorithmic_bytes", "traffic", "traffic_over_algorithmic", "l2_to_lds"):
        assert k in roof, k
    assert len(roof["kernel"]) <= 200 and "by_shape" not in roof and "kernels" not in compact
    assert roof["frac"] == full["roofline"]["frac"] and compact["cpu_baseline"] == full["cpu_baseline"]
    assert detail["kernels"] and detail["roofline"]["by_shape"]            # nothing is lost: the detail is the full report
    # emit(): one line on stdout, the report in the side file
    monkeypatch.setattr(bench, "DETAIL_PATH", str(tmp_path / "bench_detail.json"))
    bench.emit(full)
    out = capsys.readouterr()
    assert out.out.count("\n") == 1 and len(out.out) < 6000 and len(out.err) < 500
    assert json.loads(out.out)["value"] == full["value"]
    assert json.load(open(tmp_path / "bench_detail.json"))["kernels"].keys() == full["kernels"].keys()
    # a secondary-recipe line (roofline with by_shape, no kernels) and a line without a stage report
    sec = json.load(open(os.path.join(ROOT, "profiles", "r05_workload_seg.json")))
    c2, _ = bench.compact_line(sec)
    assert len(json.dumps(c2)) < 6000 and c2["roofline"]["frac"] == sec["roofline"]["frac"] and "by_shape" not in c2["roofline"]
    c3, _ = bench.compact_line({"metric": "m", "value": 1.0, "roofline": None})
    assert c3 == {"metric": "m", "value": 1.0, "roofline": None}


def test_bench_timed_repeats_times_exactly_steps_per_repeat():
    bench = _load_bench()
    calls, fences = [], []
    ts = bench.timed_repeats(lambda: calls.append(1), lambda: fences.append(len(calls)), steps=7, repeats=3)
    assert len(calls) == 21 and fences == [0, 7, 7, 14, 14, 21] and ts == sorted(ts) and len(ts) == 3
