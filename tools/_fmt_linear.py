import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        r = json.loads(l)
        tiles = sorted(((v, k[8:]) for k, v in r.items() if k.startswith("us_tile")))
        print(r["shape"], r["tile"], "ours %.1f lib %.1f gelu_d %.1f err %.1e/%.1e/%.1e |" % (r["us_ours"], r["us_lib"], r["us_ours_gelu_d"], r["err_ours"], r["err_lib"], r["err_gelu"]),
              " ".join("%s:%.1f" % (k, v) for v, k in tiles[:9]))
    elif "amdgpu.ids" not in l:
        print(l.rstrip()[:300])
