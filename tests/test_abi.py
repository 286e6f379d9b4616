"""The C-ABI library builds, loads on a GPU-less host and exports every symbol include/upp_hip.h declares."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT
from upp_hip import _abi


def _declared():
    text = open(os.path.join(ROOT, "include", "upp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(upp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_abi.LIB_PATH)
    names = _declared()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(_abi.SIGNATURES), "ctypes table and header disagree"


def test_version_and_error_strings():
    lib = _abi.load()
    assert lib.upp_abi_version() == 5 == _abi.ABI_VERSION
    assert b"null pointer" in lib.upp_error_string(-1)
    assert b"range" in lib.upp_error_string(-2)
    assert b"knn" in lib.upp_error_string(-3)
    assert lib.upp_emd_work_floats(2, 10, 20) == 2 * 30 * 11


def test_options_are_the_librarys_only_state_and_nothing_reads_the_environment():
    """ABI 5 (include/upp_hip.h "options"): four documented process-wide options behind upp_set_option / upp_get_option, the product's
    defaults, values outside the documented sets refused; no getenv left in the library (sources AND the built object's imports); the
    measured-slower experiments of round 5 are gone from the export table."""
    import subprocess
    lib = _abi.load()
    assert _abi.OPTIONS == {"SB_TUNED": 0, "SB_XCD2D": 1, "STORE_WT": 2, "EMBED_SPLIT_BF16": 3}
    hdr = open(os.path.join(ROOT, "include", "upp_hip.h")).read()
    for name, key in _abi.OPTIONS.items():
        assert re.search(r"UPP_OPT_%s = %d\b" % (name, key), hdr), name
        assert ("env UPP_" + name) in hdr
    assert lib.upp_get_option(99) == -1 and lib.upp_set_option(99, 0) == -1 and lib.upp_set_option(-1, 0) == -1
    was = [lib.upp_get_option(k) for k in range(4)]
    try:
        assert lib.upp_set_option(1, 3) == -2 and lib.upp_set_option(0, 2) == -2 and lib.upp_set_option(2, -1) == -2
        for k, vals in ((0, (0, 1)), (1, (0, 2, 4)), (2, (0, 1)), (3, (0, 1))):
            for v in vals:
                assert lib.upp_set_option(k, v) == 0 and lib.upp_get_option(k) == v
        # the option is live: the tile choice of a swept problem with and without the measured table (host functions)
        lib.upp_set_option(0, 1)
        t_on = lib.upp_linear_sb_tile(2400, 384, 1536)
        lib.upp_set_option(0, 0)
        t_off = lib.upp_linear_sb_tile(2400, 384, 1536)
        assert t_on > 0 and t_off > 0
    finally:
        for k, v in enumerate(was):
            lib.upp_set_option(k, v)
    if "UPP_SB_TUNED" not in os.environ and "UPP_SB_XCD2D" not in os.environ and "UPP_STORE_WT" not in os.environ and "UPP_EMBED_SPLIT_BF16" not in os.environ:
        assert was == [1, 2, 1, 1]                        # the product's defaults
    csrc = os.path.join(ROOT, "iccv2025-upp_amd", "upp_hip", "csrc")
    for f in os.listdir(csrc):
        assert "getenv" not in open(os.path.join(csrc, f)).read(), f
    undefined = subprocess.run(["nm", "-D", "--undefined-only", _abi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in undefined
    exported = subprocess.run(["nm", "-D", "--defined-only", _abi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    for gone in ("upp_ln_adapter_fwd_next", "upp_linear_sb_resid_f32", "upp_linear_sb_ln_f32", "upp_linear_sb_ln_usable"):
        assert gone not in exported and gone not in hdr.split("#define UPP_E_BADARG")[1]


def test_argument_validation_happens_before_any_launch():
    lib = _abi.load()
    # null pointers / bad sizes are rejected on the host: no GPU is needed to see the error code
    assert lib.upp_fps(None, None, None, 1, 8, 4, None) == -1
    assert lib.upp_knn(None, None, None, None, None, 1, 8, 4, 2, None) == -1
    assert lib.upp_chamfer_fwd(None, None, None, None, None, None, 1, 8, 8, None) == -1
    with pytest.raises(RuntimeError, match="null pointer"):
        _abi.check(-1)


def test_cpu_tensors_are_rejected_not_silently_served():
    from upp_hip import ops
    from knn_cuda import KNN
    from pointnet2_ops import pointnet2_utils
    from extensions.chamfer_dist import ChamferDistanceL1
    import emd
    x = torch.rand(2, 16, 3)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.fps(x, 4)
    with pytest.raises(RuntimeError, match="no CPU path"):
        pointnet2_utils.furthest_point_sample(x, 4)
    with pytest.raises(RuntimeError, match="no CPU path"):
        KNN(k=4, transpose_mode=True)(x, x[:, :4])       # constructible without a GPU, unusable on CPU
    with pytest.raises(RuntimeError, match="no CPU path"):
        ChamferDistanceL1()(x, x)
    with pytest.raises((RuntimeError, AssertionError)):
        emd.emd()(x, x)


def test_linear_decomposition_choice_is_a_host_function():
    """upp_linear_tile: one workgroup per CU in one round with the fewest MFMAs per SIMD (csrc/linear.hip pick_config)."""
    lib = _abi.load()

    def cfg(M, N, K):
        c = lib.upp_linear_tile(M, N, K)
        return (c >> 12) & 15, (c >> 8) & 15, (c >> 4) & 15, c & 15

    RT_TALL = 0x2000000 + 0x100000 * 5 + 0x10000 + 4096 * 2 + 256 * 2 + 16 + 1     # 2 x 2 waves of 2 x 2 blocks, 2 stages of 32 (csrc/linear_rt.hip)
    for M in (2400, 2080, 2048, 1120, 65536, 1):
        for N, K in ((1152, 384), (384, 384), (1536, 384), (384, 1536), (384, 1152), (96, 384), (40, 256)):
            bmb, bnb, ks, kc = cfg(M, N, K)
            if lib.upp_linear_tile(M, N, K) & 0x10000:          # tall: thousands of 128 x 128 tiles -> the register-tiled kernel, no split of k
                assert lib.upp_linear_tile(M, N, K) == RT_TALL and M == 65536 and N >= 256 and (ks, kc) == (1, 1)
                continue
            assert 4 <= bmb * bnb * ks <= 16 and K % (32 * ks * kc) == 0
            wgs = -(-M // (32 * bmb)) * -(-N // (32 * bnb))
            if M <= 2400:
                assert wgs <= 256, (M, N, K, wgs)
    assert cfg(2400, 1536, 384)[:3] == (4, 4, 1)           # 228 workgroups of 16 blocks: 4 per SIMD
    assert cfg(2048, 1536, 384)[:3] == (4, 3, 1)           # 256 workgroups of 12 blocks: 3 per SIMD
    assert cfg(2400, 384, 1536)[:3] == (2, 2, 4)           # 228 workgroups, contraction split 4 ways: 1 block per SIMD
    assert cfg(2400, 384, 384)[:3] == (2, 2, 4)            # (the 2-way split is faster stand-alone and slower in the two-stream step)
    assert lib.upp_linear_tile(0, 1, 32) == -1 and lib.upp_linear_tile(32, 32, 50) == -2 and lib.upp_linear_tile(32, 32, 48) > 0
    # problems that fit no one-round decomposition (part segmentation: 4,096 ... 4,448 token rows): the fitted multi-round model picks
    # within 2 % of the best measured choice (profiles/r03_time_linear_rows.jsonl) -- (2,2,2,64) or the register-tiled kernel for the wide
    # layers, (2,4,2,64) for the narrow ones; the one-round model's leftover (1,2,4,128) cost fc1 47 % there
    RT = 0x2512211
    assert lib.upp_linear_tile(4416, 1536, 384) in (0x2221, RT) and lib.upp_linear_tile(4128, 1152, 384) in (0x2221, RT)
    assert lib.upp_linear_tile(4416, 384, 1536) == 0x2421 and lib.upp_linear_tile(4128, 384, 1152) == 0x2421
    assert lib.upp_linear_tile(2720, 1536, 384) == 0x2321 and lib.upp_linear_tile(6144, 1152, 384) == RT
    assert lib.upp_linear_tile(65536, 1024, 1536) == RT and lib.upp_linear_tile(65536, 512, 256) == RT


def _sb_compiled_and_rows():
    import re
    csrc = os.path.join(ROOT, "iccv2025-upp_amd", "upp_hip", "csrc")
    compiled = set()
    for f, macro in (("linear_sb.hip", "UPP_SB_CONFIGS"), ("linear_sb_tuned.h", "UPP_SB_TUNED_CONFIGS")):
        m = re.search(r'#define %s\(X\) (.*)\n' % macro, open(os.path.join(csrc, f)).read())
        for t in re.findall(r'X\(([^)]*)\)', m.group(1).replace('UPP_SB_NST44', '3')):
            a, b, c, d, e = (int(v) for v in t.split(','))
            compiled.add(0x400000 + a * 65536 + b * 4096 + c * 256 + d * 16 + e)
    rows = [(int(a), int(b), int(c), int(d, 16)) for a, b, c, d in
            re.findall(r'\{(\d+), (\d+), (\d+), 0x([0-9a-f]+)\}', open(os.path.join(csrc, "linear_sb_tuned.h")).read())]
    return compiled, rows


def test_split_bf16_tuned_tile_table_is_well_formed_and_answers_for_its_own_problems():
    """csrc/linear_sb_tuned.h (round 6: the RESIDUAL of the fitted cost model, generated by tools/micro/sb_model_fit.py): fewer than 30
    rows; every row names a compiled tile shape whose wave groups' k-stages divide K and fill its LDS stages, upp_linear_sb_tile (a host
    function) answers with the row's tile for the row's problem (exact matches only: a neighbouring problem takes the model's choice)."""
    if _abi.load().upp_get_option(_abi.OPTIONS["SB_TUNED"]) == 0:
        pytest.skip("option SB_TUNED = 0: the library answers with the cost model alone")
    compiled, rows = _sb_compiled_and_rows()
    assert 1 <= len(rows) < 30
    lib = _abi.load()
    for M, N, K, code in rows:
        bmb, bnb, ks, nst = (code >> 16) & 15, (code >> 12) & 15, (code >> 4) & 15, code & 15
        assert code in compiled, hex(code)
        assert K % (32 * ks) == 0 and K // (32 * ks) >= nst, (M, N, K, hex(code))
        assert lib.upp_linear_sb_tile(M, N, K) == code, (M, N, K, hex(code))
    # far from every swept M, and shapes of no recipe: the cost model answers with a compiled tile that serves the problem
    for M, N, K in ((300, 384, 384), (1800, 1152, 384), (3600, 384, 1536), (5000, 640, 320), (77, 96, 64), (100000, 2048, 4096)):
        t = lib.upp_linear_sb_tile(M, N, K)
        assert t in compiled and K % (32 * ((t >> 4) & 15)) == 0 and K // (32 * ((t >> 4) & 15)) >= (t & 15), (M, N, K, hex(t))
    assert lib.upp_linear_sb_tile(64, 64, 48) == 0 and lib.upp_linear_sb_tile(64, 64, 32) == 0      # K % 32, K < 64: not this kernel's


def test_split_bf16_cost_model_picks_within_3_percent_of_the_measured_best():
    """Round 5's verdict: the analytic model behind the table chose within 3 % of the measured best on 23 of the 197 swept problems.  The
    model is now fitted to that sweep (profiles/r05_sb_sweep.json; per compiled tile seven coefficients in rounds, k-stages and chip fill:
    csrc/linear_sb_model.h).  Asserted on the library's own answers (host code): with the table OFF the choice is within 3 % of the best
    compiled tile on >= 90 % of the swept problems and costs < 1.5 % of summed time; with the residual table on, every swept problem is
    within 3.5 %.  Generalisation to problems outside the sweep: tools/micro/sb_model_fit.py (cross-validation) and
    profiles/r06_sb_model_b24_b48.txt (a new sweep at B = 24 and 48, measured)."""
    import json
    lib = _abi.load()
    compiled, _ = _sb_compiled_and_rows()
    sweep = json.load(open(os.path.join(ROOT, "profiles", "r05_sb_sweep.json")))
    key = _abi.OPTIONS["SB_TUNED"]
    was = lib.upp_get_option(key)

    def regrets():
        out, tot_pick, tot_best = [], 0.0, 0.0
        for r in sweep:
            cand = {int(c, 16): us for c, us in r["us"].items()
                    if int(c, 16) in compiled and r["K"] % (32 * ((int(c, 16) >> 4) & 15)) == 0 and r["K"] // (32 * ((int(c, 16) >> 4) & 15)) >= (int(c, 16) & 15)}
            pick = lib.upp_linear_sb_tile(r["M"], r["N"], r["K"])
            assert pick in cand, (r["M"], r["N"], r["K"], hex(pick))
            best = min(cand.values())
            out.append(cand[pick] / best - 1.0)
            tot_pick += cand[pick]; tot_best += best
        return out, tot_pick / tot_best - 1.0
    try:
        lib.upp_set_option(key, 0)
        reg, extra = regrets()
        assert len(reg) == 197
        assert sum(r <= 0.03 for r in reg) >= 0.90 * len(reg), sum(r <= 0.03 for r in reg)
        assert extra < 0.015 and max(reg) < 0.12, (extra, max(reg))
        lib.upp_set_option(key, 1)
        reg, extra = regrets()
        assert max(reg) <= 0.035 and extra < 0.01, (max(reg), extra)
    finally:
        lib.upp_set_option(key, was)


def test_exported_symbols_are_exactly_the_declared_ones():
    """nm -D libupp_hip.so | grep ' T upp_'  ==  the prototypes of include/upp_hip.h: no undeclared hooks, no global switches."""
    import subprocess
    out = subprocess.check_output(["nm", "-D", _abi.LIB_PATH]).decode()
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T upp_" in l)
    assert exported == _declared()


def _gfx950_disassembly():
    """Disassembly text of every gfx950 code object inside libupp_hip.so (llvm-objdump -d)."""
    import re
    import struct
    import subprocess
    import tempfile
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("no llvm-objdump")
    from upp_hip import build as B
    data = open(os.path.join(os.path.dirname(B.__file__), "lib", "libupp_hip.so"), "rb").read()
    texts = []
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data):
        p = m.start()
        (n,) = struct.unpack_from("<Q", data, p + 24)
        q = p + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, q)
            q += 24
            triple = data[q:q + tl].decode()
            q += tl
            if "gfx950" not in triple or size == 0:
                continue
            with tempfile.NamedTemporaryFile(suffix=".co") as f:
                f.write(data[p + off:p + off + size])
                f.flush()
                texts.append(subprocess.run([objdump, "-d", f.name], capture_output=True, text=True, check=True).stdout)
    return texts


def test_only_the_listed_kernels_touch_scratch_memory():
    """Round 6: a lambda that captured three float4 staging registers by reference and selected between whole vectors put them into
    scratch memory -- 80 bytes of private segment that cost adapter_wgrad_kernel 31 of its 46 us (NOTEBOOK 12.10).  No error, no warning.
    Here: every kernel of the library is disassembled; `scratch_` instructions may appear only in the two kernels that are known to
    spill a few registers at their 256-register limit (attn_bwd_long_kernel: 4-12 VGPRs in a depth-1 loop) or to index a per-lane
    array (fps_kernel with 64 points per lane: the large-cloud form of the batch preparation)."""
    import re
    allowed = ("attn_bwd_long_kernel", "fps_kernelILi64")
    offenders, kernels = {}, 0
    for text in _gfx950_disassembly():
        name = None
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                name = m.group(1)
                kernels += 1
            elif name and "scratch_" in line and not any(a in name for a in allowed):
                offenders[name] = offenders.get(name, 0) + 1
    assert kernels > 200
    assert not offenders, "kernels with scratch accesses: %s" % sorted(offenders.items())


def test_the_library_only_launches_kernels():
    """Every entry point may be captured into a HIP graph, and a memset node of a graph replays correctly ONCE on this stack (ROCm 7.2.0:
    NOTEBOOK 12.11) -- so the library must not memset, memcpy, allocate or synchronise: the runtime symbols it imports are the launch,
    the dynamic-LDS attribute and the two error queries."""
    import subprocess
    out = subprocess.check_output(["nm", "-D", _abi.LIB_PATH]).decode()
    imported = sorted({l.split()[-1].split("@")[0] for l in out.splitlines() if " U hip" in l})
    assert imported == ["hipFuncSetAttribute", "hipGetErrorString", "hipGetLastError", "hipLaunchKernel"], imported


def test_library_contains_no_packed_f32_instruction():
    """Round 4: v_pk_add_f32 with op_sel:[0,1] returns a - 0 in its low half every so often while a bf16-MFMA workgroup shares the CU
    (tools/micro/src/lds_canary.cpp; it changed FPS picks in the pipelined step), so the library is built with -target-feature
    -packed-fp32-ops (upp_hip/build.py).  Here: every gfx950 code object inside libupp_hip.so is disassembled and must hold kernels,
    matrix instructions and NO v_pk_{add,mul,fma}_f32."""
    import re
    import struct
    import subprocess
    import tempfile
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("no llvm-objdump")
    from upp_hip import build as B
    data = open(os.path.join(os.path.dirname(B.__file__), "lib", "libupp_hip.so"), "rb").read()
    packed = mfma = kernels = 0
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data):
        p = m.start()
        (n,) = struct.unpack_from("<Q", data, p + 24)
        q = p + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, q)
            q += 24
            triple = data[q:q + tl].decode()
            q += tl
            if "gfx950" not in triple or size == 0:
                continue
            with tempfile.NamedTemporaryFile(suffix=".co") as f:
                f.write(data[p + off:p + off + size])
                f.flush()
                text = subprocess.run([objdump, "-d", f.name], capture_output=True, text=True, check=True).stdout
            packed += len(re.findall(r"v_pk_(?:add|mul|fma)_f32", text))
            mfma += text.count("v_mfma")
            kernels += text.count("s_endpgm")
    assert kernels > 200 and mfma > 1000, (kernels, mfma)
    assert packed == 0, "%d packed-f32 instructions in libupp_hip.so" % packed


def test_wgrad_sb_row_plan_is_well_formed_on_the_host():
    """upp_linear_wgrad_grouped_sb_rows (host-only planning, no GPU call): every run length is a whole multiple of 32 rows, at most the
    problem's rows rounded up to 32, the same plan for the same group, and tall problems are cut into many runs while short ones are not."""
    lib = _abi.load()
    groups = [
        [(1216, 96, 384)] + [(2048, 1536, 384), (2048, 384, 1536), (2048, 384, 384), (2048, 1152, 384)] * 4
        + [(832, 1536, 384), (832, 384, 1536), (832, 384, 384), (832, 1152, 384)] * 12 + [(65536, 384, 512), (65536, 256, 128)],
        [(65536, 256, 512), (65536, 1024, 1536), (32, 128, 64), (32, 64, 16)],
        [(1, 4, 4)], [(33, 36, 260)], [(140000, 256, 260)],
    ]
    for grp in groups:
        k = len(grp)
        M = (ctypes.c_int * k)(*[g[0] for g in grp]); N = (ctypes.c_int * k)(*[g[1] for g in grp]); K = (ctypes.c_int * k)(*[g[2] for g in grp])
        r1, r2 = (ctypes.c_int * k)(), (ctypes.c_int * k)()
        assert lib.upp_linear_wgrad_grouped_sb_rows(k, M, N, K, r1) == 0 and lib.upp_linear_wgrad_grouped_sb_rows(k, M, N, K, r2) == 0
        assert list(r1) == list(r2)
        for (m, n, kk), rows in zip(grp, r1):
            assert rows >= 32 and rows % 32 == 0 and rows <= (m + 31) // 32 * 32, (m, n, kk, rows)
            if m >= 65536 and n > 128 and kk > 128:
                assert (m + rows - 1) // rows >= 8, (m, n, kk, rows)          # a tall problem feeds many workgroups
            if m <= 1024:
                assert (m + rows - 1) // rows <= 4, (m, n, kk, rows)
    assert lib.upp_linear_wgrad_grouped_sb_rows(0, None, None, None, None) != 0
