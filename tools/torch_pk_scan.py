"""Packed-f32 exposure of torch's OWN kernels (round-4 finding: v_pk_{add,mul,fma}_f32 whose op_sel routes the HIGH half of src1 to the low
lane returned wrong low halves beside a bf16-MFMA workgroup; libupp_hip.so is built without packed f32, torch's element-wise kernels are not
ours to rebuild).  This tool unbundles every gfx950 code object of libtorch_hip.so (compressed offload bundles: clang-offload-bundler),
disassembles it and reports, per kernel whose demangled name contains one of the given substrings, the packed-f32 instructions it holds
and how many of them carry the affected operand routing (an `op_sel:[x,1...]` whose second entry is 1: src1's high half feeds the low lane).

    python tools/torch_pk_scan.py [substring ...] > profiles/r05_torch_pk_scan.txt        (CPU only; several minutes)

Default substrings: the torch kernels the headline / segmentation steps still launch (profiles/r0N_glue_census*.txt)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
DEFAULT = ["CUDAFunctor_add", "FillFunctor", "CatArrayBatchedCopy", "vectorized_layer_norm_kernel", "MaxOps<float>", "sum_functor", "vectorized_gather_kernel",
           "direct_copy_kernel", "AUnaryFunctor<float, float, float, at::native::binary_internal::MulFunctor", "BinaryFunctor<float, float, float, at::native::binary_internal::MulFunctor",
           "multi_tensor_apply_kernel<at::native::TensorListScalarListMetadata<long, 1>", "uniform_and_transform", "BUnaryFunctor<float, float, float, at::native::binary_internal::MulFunctor"]


def main():
    import torch
    lib = os.path.join(os.path.dirname(torch.__file__), "lib", "libtorch_hip.so")
    wanted = sys.argv[1:] or DEFAULT
    work = tempfile.mkdtemp(prefix="torch_pk_")
    fat = os.path.join(work, "fat.bin")
    subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, lib, os.path.join(work, "ignore.so")], check=True)
    os.remove(os.path.join(work, "ignore.so"))
    data = open(fat, "rb").read()
    import struct
    spans = []
    for m in re.finditer(b"CCOB", data):         # compressed offload bundle, version 2: magic, u16 version, u16 method, u32 FILE SIZE, u32 raw size, u64 hash
        a = m.start()
        ver, _method, fsize = struct.unpack_from("<HHI", data, a + 4)
        if ver >= 2 and 24 < fsize <= len(data) - a:
            spans.append((a, a + fsize))
    per_kernel = collections.OrderedDict()
    n_obj = n_kern = total_pk = total_bad = 0
    for a, b in spans:
        chunk, co = os.path.join(work, "chunk"), os.path.join(work, "co")
        open(chunk, "wb").write(data[a:b])
        r = subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + chunk,
                            "--output=" + co], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
            continue
        n_obj += 1
        p = subprocess.Popen([LLVM + "/llvm-objdump", "-d", "--demangle", co], stdout=subprocess.PIPE, text=True)
        name = None
        for line in p.stdout:
            m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
            if m:
                name = m.group(1)
                n_kern += 1
                continue
            if "v_pk_" in line and re.search(r"v_pk_(add|mul|fma)_f32", line):
                total_pk += 1
                ops = re.search(r"op_sel:\[([01,]+)\]", line)
                bad = bool(ops and len(ops.group(1).split(",")) > 1 and ops.group(1).split(",")[1] == "1")
                total_bad += bad
                if name and any(w in name for w in wanted):
                    e = per_kernel.setdefault(name, [0, 0])
                    e[0] += 1
                    e[1] += bad
            elif name and any(w in name for w in wanted):
                per_kernel.setdefault(name, [0, 0])
        p.wait()
        os.remove(co)
    print("libtorch_hip.so: %d gfx950 code objects, %d functions; v_pk_{add,mul,fma}_f32: %d in all, %d with src1's high half routed to the low lane (op_sel:[x,1,..])"
          % (n_obj, n_kern, total_pk, total_bad))
    print("kernels matching %s:" % wanted)
    groups = collections.OrderedDict()
    for k, (pk, bad) in per_kernel.items():
        key = next(w for w in wanted if w in k)
        g = groups.setdefault(key, [0, 0, 0, 0])
        g[0] += 1; g[1] += pk; g[2] += bad; g[3] += pk > 0
    for key, (n, pk, bad, with_pk) in groups.items():
        print("  %-90s %5d instantiations, %4d of them with packed f32 (%5d instructions), affected routing: %d" % (key[:90], n, with_pk, pk, bad))
    worst = sorted(((bad, pk, k) for k, (pk, bad) in per_kernel.items() if bad), reverse=True)[:40]
    if worst:
        print("instantiations with the affected routing:")
        for bad, pk, k in worst:
            print("  %4d / %4d  %s" % (bad, pk, k[:300]))


if __name__ == "__main__":
    main()
