import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xgpr_amd import xgpr_hip_rfgen_ext as ext
from oracle import oracle as orc
orc.build(ref=False); oracle = orc.Oracle()
dev = "cuda"
def T(a): return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
def run(C, cw, L, m2, sl, sc, seed=0, tag=""):
    rng = np.random.default_rng(seed)
    ns = len(sl)
    radem2, chi2 = orc.draw_sorf_params(m2, cw * C, 77, conv=True)
    xs = rng.standard_normal((ns, L, C)).astype(np.float32)
    sl = np.asarray(sl, dtype=np.int32)
    refc = np.zeros((ns, m2)); oracle.cpuConv1dFGen(xs, refc, radem2, chi2, sl, cw, sc)
    oc = torch.zeros((ns, m2), dtype=torch.float64, device=dev)
    ext.hipConv1dFGen(T(xs), oc, T(radem2), T(chi2), sl, cw, sc)
    err = np.abs(oc.cpu().numpy() - refc)
    kmax = int(sl.max()) - cw + 1
    cscale = np.sqrt(2.0 / m2) * {0: kmax, 1: np.sqrt(kmax), 2: 1.0}[sc]
    bad = np.argwhere(err > 4e-7 * cscale)
    print(f"{tag} C={C} w={cw} L={L} M={m2} sl={sl.tolist()} sc={sc} radem{radem2.shape}: max err {err.max():.3e} (bar {4e-7*cscale:.1e}) bad entries {len(bad)}",
          ("rows %s cols %d..%d" % (sorted(set(bad[:,0].tolist())), bad[:,1].min(), bad[:,1].max())) if len(bad) else "")
run(64, 8, 36, 600, [14, 29, 17, 20], 2, tag="orig")
for sc in (0, 1, 2): run(64, 8, 36, 600, [14, 29, 17, 20], sc, tag="sc")
for m2 in (64, 512, 600, 1024, 1026, 2048): run(64, 8, 36, m2, [14, 29, 17, 20], 2, tag="M")
run(64, 8, 36, 600, [20], 2, tag="1seq"); run(64, 8, 36, 600, [8], 2, tag="1kmer"); run(64, 8, 36, 600, [36]*4, 2, tag="full")
run(32, 16, 36, 600, [20, 30], 2, tag="C32w16"); run(64, 4, 36, 600, [20, 30], 2, tag="P256"); run(64, 16, 36, 600, [20, 30], 2, tag="P1024"); run(21, 9, 36, 600, [20, 30], 2, tag="P256pad")
run(64, 8, 36, 600, [14, 29, 17, 20], 2, seed=5, tag="seed5")
