#!/usr/bin/env python3
"""HBM-side traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE need separate passes: 3 + 2 of the
4 TCC slots, MI355X_MICROARCH.md 'rocprofv3 PMC slots').

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out_f -o f -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out_w -o w -- python3 bench.py ... (same command)
  python tools/pmc_traffic.py out_f/f_counter_collection.csv out_w/w_counter_collection.csv profiles/r01_c2_pmc_traffic.json

Units and corrections (same guide, 'HBM'): both counters are reported in KiB; on gfx950 FETCH_SIZE tallies the 128-byte
requests of wide coalesced reads at 64 bytes, so fetched bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is exact for 16-byte
streaming stores.  Infinity-Cache hits are included in both (fabric-side counters)."""
import csv
import json
import re
import sys
from collections import defaultdict

# GPU kernels behind each C-ABI entry whose kernels are uniquely named (bench.py's roofline classes)
CLASSES = {
    "dy_conv2d_wgrad": [r"wg2::wgrad_kernel", r"wg2::reduce_kernel", r"wg3::wgrad_kernel", r"wg3::reduce_kernel", r"wg4::wgrad_kernel",
                        r"wg4::reduce_kernel", r"conv_wgrad_kernel", r"wgrad_reduce_kernel", r"dense_wgrad_kernel"],
    # forward and input-gradient convs run the same kernels (a dgrad is a conv over the transposed, flipped weights)
    "dy_conv2d_fwd+dy_conv2d_dgrad": [r"v4::conv_kernel", r"v5::conv_kernel", r"v5::band_kernel", r"v3::conv3x3_kernel", r"v2::conv_kernel", r"conv_igemm_kernel",
                                      r"conv_thin_kernel", r"dgrad1x1_thin_kernel", r"dgrad3x3s2_small_kernel", r"conv_dense_kernel",
                                      r"conv_generic_kernel", r"conv_small_kernel"],
    "dy_bn_act_bwd_reduce": [r"bn_act_bwd_reduce_kernel"],
    "dy_bn_act_bwd_apply": [r"bn_act_bwd_apply_kernel"],
    "dy_bn_act_fwd": [r"bn_act_fwd_kernel"],
    "dy_usm_bwd": [r"usm_bwd_kernel"],
}
MAIN = {"dy_conv2d_wgrad": [r"wg2::wgrad_kernel", r"wg3::wgrad_kernel", r"wg4::wgrad_kernel", r"conv_wgrad_kernel", r"dense_wgrad_kernel"]}      # one dispatch of these per C-ABI call


def load(path, counter):
    agg = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = agg[r["Kernel_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    return agg


_DEMANGLE = None


def demangle(name):
    """rocprofv3 leaves the _Float16 instantiations mangled (its demangler does not know DF16_); libstdc++'s does once the token is
    spelled as the Itanium `half` (Dh): `v4::conv_kernel<0, half>` -> written back as `_Float16`."""
    global _DEMANGLE
    if not name.startswith("_Z"):
        return name
    if _DEMANGLE is None:
        import ctypes
        lib, libc = ctypes.CDLL("libstdc++.so.6"), ctypes.CDLL("libc.so.6")
        lib.__cxa_demangle.restype = ctypes.c_void_p

        def run(n):
            st = ctypes.c_int(0)
            p = lib.__cxa_demangle(n.encode(), None, None, ctypes.byref(st))
            if not p:
                return None
            out = ctypes.string_at(p).decode()
            libc.free(ctypes.c_void_p(p))
            return out
        _DEMANGLE = run
    out = _DEMANGLE(name.replace("DF16_", "Dh"))
    return name if out is None else re.sub(r"\bhalf\b", "_Float16", out)


def short(name):
    name = demangle(name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"^void ", "", name).split("(")[0]


def main():
    f, w, out = sys.argv[1:4]
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5          # steps the profiled command ran (warm-up included)
    fa, wa = load(f, "FETCH_SIZE"), load(w, "WRITE_SIZE")
    kernels = {}
    for k in set(fa) | set(wa):
        fk, wk = fa.get(k, [0.0, 0]), wa.get(k, [0.0, 0])
        n = max(fk[1], wk[1])
        kernels[short(k)] = dict(launches=n, fetch_bytes_corrected=2.0 * fk[0] * 1024, write_bytes=wk[0] * 1024,
                                 fetch_size_raw_kib=fk[0], write_size_raw_kib=wk[0])
    classes = {}
    for cname, pats in CLASSES.items():
        sel = [v for k, v in kernels.items() if any(re.search(p, k) for p in pats)]
        if not sel:
            continue
        main_pats = MAIN.get(cname, pats)
        calls = sum(v["launches"] for k, v in kernels.items() if any(re.search(p, k) for p in main_pats))
        tot = sum(v["fetch_bytes_corrected"] + v["write_bytes"] for v in sel)
        classes[cname] = dict(calls=calls, traffic_bytes_total=tot, traffic_bytes_per_call=tot / max(calls, 1))
    json.dump(dict(note="traffic = 2*FETCH_SIZE + WRITE_SIZE (KiB -> bytes), per MI355X_MICROARCH.md HBM section", steps=steps, classes=classes,
                   kernels=dict(sorted(kernels.items(), key=lambda kv: -(kv[1]["fetch_bytes_corrected"] + kv[1]["write_bytes"])))),
              open(out, "w"), indent=1)
    for c, v in classes.items():
        print(f"{c:24s} calls {v['calls']:6d}  traffic/call {v['traffic_bytes_per_call'] / 1e6:9.2f} MB")


if __name__ == "__main__":
    main()
