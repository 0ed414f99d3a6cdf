"""HBM traffic per kernel launch from two rocprofv3 --pmc passes of the bench command (MI355X_MICROARCH.md, section HBM):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d <out>/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d <out>/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
    python tools/hbm_traffic.py <out>/fetch <out>/write profiles/r03_hbm_traffic.json

FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes, so reads are
doubled (the guide's correction for wide coalesced reads); WRITE_SIZE is exact for 16-byte stores and float atomics.
Kernels are grouped into the families bench.py reports (same template -> same family)."""
import collections, csv, glob, json, sys

# (family, kernel-name substrings that must ALL occur): the 128x128 templates bench.py names in KERNEL_OF
# (rocprofv3's demangler garbles the TB = true instantiations of gemm256_kernel into "gemm256_kernel<bool _Accum, bool, E, ...>":
# that spelling therefore identifies the NN products)
FAMILY = [("wgrad_group", ("wgrad_group_kernel",)), ("gemm256_nn", ("gemm256_kernel", "_Accum")),
          ("gemm256_nt", ("gemm256_kernel", "Lb0E")), ("gemm256_nn", ("gemm256_kernel", "Lb1E")),
          ("gemm256_nt", ("gemm256_kernel", "false")), ("gemm256_nn", ("gemm256_kernel", "true")),
          ("gemm_tn", ("gemm_tn2_kernel",)), ("gemm_tn", ("gemm_fast_kernel", "true, true, 128, 128")), ("gemm_tn", ("gemm_fast_kernel", "Lb1ELb1ELi128ELi128")),
          ("gemm_nt", ("gemm_fast_kernel", "Lb0ELb0ELi128ELi128")), ("gemm_nn", ("gemm_fast_kernel", "Lb0ELb1ELi128ELi128")),
          ("gemm_nt", ("gemm_fast_kernel", "false, false, 128, 128")), ("gemm_nn", ("gemm_fast_kernel", "false, true, 128, 128")),
          ("gemm_gather", ("gemm_kernel",)), ("conv2_fwd", ("conv2_fwd_kernel",)), ("conv2_dgrad", ("conv2_dgrad_kernel",)), ("attn_fwd", ("attn_fwd2",)), ("attn_bwd", ("attn_bwd_d",))]


def family_of(name):
    for fam, subs in FAMILY:
        if all(s in name for s in subs):
            return fam
    return None


def load(d, counter):
    per = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return per


def main():
    fetch, write, out = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE"), sys.argv[3]
    fams = collections.defaultdict(lambda: dict(launches=0, read=0.0, written=0.0))
    kernels = {}
    for name in set(fetch) | set(write):
        rd = sum(fetch.get(name, [])) * 1024.0 * 2.0          # KiB -> bytes, gfx950 x2
        wr = sum(write.get(name, [])) * 1024.0
        n = max(len(fetch.get(name, [])), len(write.get(name, [])), 1)
        kernels[name[:120]] = dict(launches=n, read_bytes_per_launch=rd / n, written_bytes_per_launch=wr / n)
        fam = family_of(name)
        if fam:
            fams[fam]["launches"] += n; fams[fam]["read"] += rd; fams[fam]["written"] += wr
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py; reads x2 (gfx950), KiB units",
           "families": {f: dict(launches=v["launches"], read_bytes_per_launch=round(v["read"] / v["launches"]),
                                written_bytes_per_launch=round(v["written"] / v["launches"]),
                                bytes_per_launch=round((v["read"] + v["written"]) / v["launches"])) for f, v in fams.items()},
           "top_kernels": dict(sorted(kernels.items(), key=lambda kv: -(kv[1]["read_bytes_per_launch"] + kv[1]["written_bytes_per_launch"]) * kv[1]["launches"])[:25])}
    json.dump(res, open(out, "w"), indent=1)
    for f, v in res["families"].items():
        print(f, v)


if __name__ == "__main__":
    main()
