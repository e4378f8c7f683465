"""bench.py --batch 256 inputs (4 slots): per-utterance error of the chunked path WITHOUT the conditioning guard
(run with GOLF_SS_PHI_GUARD=0) and of the serial kernels, against the float64 oracle; lists the rows that differ most."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import golf_oracle as O
from golf_amd import functional as GF
from golf_amd.synthetic import make_inputs
d = lambda x: torch.as_tensor(x).cuda()
for seed in range(2434, 2438):
    inp = make_inputs(B=256, seed=seed)
    ex, gain, a, hop = inp["noise"], inp["gain"], inp["a"], inp["hop"]
    yc = GF.ltv_allpole_ss(ex.cuda(), gain.cuda(), a.cuda(), hop).cpu().numpy()
    ys = GF.ltv_allpole_ss(ex.cuda(), gain.cuda(), a.cuda(), hop, mode="serial").cpu().numpy()
    dev = np.abs(yc - ys).max(1) / (np.abs(ys).max(1) + 1e-30)          # chunked vs sequential, no oracle needed
    worst = np.argsort(dev)[-4:]
    print(f"seed {seed}: rows where chunked deviates most from the sequential kernel: {worst} {dev[worst]}")
    sub = worst
    ref = O.ltv_allpole_ss_forward(ex.numpy()[sub], gain.numpy()[sub], a.numpy()[sub], hop)
    sc = np.abs(ref).max(1)
    print("    vs oracle: chunked", np.abs(yc[sub] - ref).max(1) / sc, "serial", np.abs(ys[sub] - ref).max(1) / sc, "|y|max", sc)
