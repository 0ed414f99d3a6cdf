"""GPU-side durations of the decoder-sized products (the host cannot launch them as fast as they run, so events around back-to-back
launches measure the host): run under `rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/small_gemm_trace.py run [M ...]`,
then `python tools/small_gemm_trace.py parse DIR [M ...]` prints the median duration of each case's launches (the profiler adds a
constant ~1-3 us to short kernels: compare cases, not absolutes)."""
import os, sys, glob, csv, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REP = 24

def cases(Ms):
    out = []
    for M in Ms:
        for N, Kd in ((512, 512), (512, 1536), (512, 2048), (1536, 512), (2048, 512), (512, 128), (512, 1024)):
            out.append(("NT", M, N, Kd))
            out.append(("NN", M, N, Kd))
    return out

def run(Ms):
    import torch
    from fbk_fairseq_st_amd import kernels as K
    for kv in os.environ.get("S2T_OPTS", "").split(","):          # e.g. S2T_OPTS=gemm_kgroups=2,gemm_small_nn=200
        if kv: K.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    worst = 0.0
    g = torch.Generator(device="cuda").manual_seed(0)
    def t(x, *s): return (torch.randn(*s, device="cuda", generator=g) * x).to(torch.bfloat16)
    for kind, M, N, Kd in cases(Ms):
        x = t(1, M, Kd); w = t(Kd ** -0.5, N, Kd) if kind == "NT" else t(Kd ** -0.5, Kd, N)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for _ in range(REP): K.gemm(x, w, trans_b=(kind == "NN"), out=out)
        torch.cuda.synchronize()
        ref = x.float() @ (w.float().t() if kind == "NT" else w.float())
        worst = max(worst, ((out.float() - ref).abs().max() / ref.abs().max()).item())
    print("worst relative error against f32 math: %.3g" % worst, file=sys.stderr)

def parse(d, Ms):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows = [r for r in rows if "gemm" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    cs = cases(Ms)
    assert len(rows) == REP * len(cs), (len(rows), REP * len(cs))
    for i, (kind, M, N, Kd) in enumerate(cs):
        ch = rows[i * REP:(i + 1) * REP][4:]
        dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in ch]
        name = ch[0]["Kernel_Name"]
        short = name[name.find("gemm"):][:60]
        print("%s M=%-5d N=%-5d K=%-5d  %6.1f us (min %5.1f)  %6.0f TF/s  wg %s  %s" % (
            kind, M, N, Kd, statistics.median(dur), min(dur), 2.0 * M * N * Kd / statistics.median(dur) / 1e6,
            ch[0].get("Grid_Size_X", "?") + "/" + ch[0].get("Workgroup_Size_X", "?"), short))

if __name__ == "__main__":
    Ms = [int(a) for a in sys.argv[3 if sys.argv[1] == "parse" else 2:]] or [2560]
    if sys.argv[1] == "run": run(Ms)
    else: parse(sys.argv[2], Ms)
