"""Holds the gemm256 epilogue's lane-turned stores to a twin library built with -DS2T_NOTURN (stores in accumulator order): every epilogue
variant over many shapes and repeated launches must agree BIT FOR BIT (same arithmetic, different store path).
  make -C fbk_fairseq_st_amd/csrc noturn
  python tools/gemm_turn_check.py            (runs itself once per library and compares checksums + full tensors through files)"""
import os, sys, subprocess, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [(24000, 384, 192), (24000, 512, 512), (24000, 1536, 512), (24000, 2048, 512), (24000, 512, 2048), (23000, 640, 1280),
          (6211, 1536, 512), (36800, 256, 512), (24000, 2048, 128), (12000, 1024, 1024)]
def run():
    import torch
    from fbk_fairseq_st_amd import kernels as K
    dev = "cuda"; dt = torch.bfloat16
    sums = []
    for (M, N, Kd) in SHAPES:
        g = torch.Generator(device=dev).manual_seed(M + N + Kd)
        a = torch.randn(M, Kd, device=dev, generator=g).to(dt); w = (torch.randn(N, Kd, device=dev, generator=g) * Kd ** -0.5).to(dt)
        b = torch.randn(N, device=dev, generator=g); r = torch.randn(M, N, device=dev, generator=g).to(dt)
        dy = torch.randn(M, N, device=dev, generator=g).to(dt); aux = torch.randn(M, Kd, device=dev, generator=g).to(dt)
        for rep in range(6):
            outs = [K.gemm(a, w, bias=b), K.gemm(a, w, bias=b, p_drop=0.25, seed=9 + rep), K.gemm(a, w, bias=b, residual=r, p_drop=0.1, seed=3),
                    K.gemm(a, w, bias=b, act=K.ACT_RELU, p_drop=0.1, seed=5), K.gemm(a, w, bias=b, act=K.ACT_RELU, residual=r),
                    K.gemm(dy, w, trans_b=True), K.gemm(dy, w, trans_b=True, act=K.ACT_RELU_BWD, aux=aux, alpha=1.25)]
            torch.cuda.synchronize()
            for o in outs:
                sums.append(hashlib.md5(o.view(torch.int16).cpu().numpy().tobytes()).hexdigest())
    print("\n".join(sums))
if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "run":
        run(); sys.exit(0)
    res = {}
    for name, lib in (("turn", os.path.join(ROOT, "fbk_fairseq_st_amd", "libs2t_hip.so")), ("noturn", os.path.join(ROOT, "fbk_fairseq_st_amd", "libs2t_hip_noturn.so"))):
        env = dict(os.environ, S2T_HIP_LIB=lib)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "run"], env=env, capture_output=True, text=True)
        res[name] = [l for l in out.stdout.split("\n") if len(l) == 32]
        if not res[name]: print(out.stderr[-2000:])
    n = len(res["turn"]); bad = [i for i in range(min(n, len(res["noturn"]))) if res["turn"][i] != res["noturn"][i]]
    print("%d outputs compared, %d differ%s" % (n, len(bad), (": " + str(bad[:20])) if bad else ""))
    sys.exit(1 if bad or n == 0 or n != len(res["noturn"]) else 0)
