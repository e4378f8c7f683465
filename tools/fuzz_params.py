"""dev: the parameters (B, F, M, hop, sigma, inner seed) of case k of `tools/fuzz_tiers.py cases seed`, without running anything
(replays the generator's draws; assumes every third case drew its gy, which holds unless no row of it was resolvable).
python tools/fuzz_params.py seed case [case ...]"""
import sys
import numpy as np

seed, want = int(sys.argv[1]), [int(v) for v in sys.argv[2:]]
rng = np.random.default_rng(seed)
for case in range(max(want) + 1):
    B = int(rng.integers(1, 14))
    M = int(rng.choice([6, 12, 16, 20, 22]))
    hop = int(rng.choice([240, 240, 240, 120, 480]))
    F = int(rng.integers(50, 230)) if hop != 480 else int(rng.integers(30, 120))
    sigma = float(rng.choice([0.3, 0.7, 1.0, 1.3]))
    inner = int(rng.integers(1 << 30))
    if case in want:
        print(f"seed {seed} case {case}: (B, F, M, hop, sigma, inner) = ({B}, {F}, {M}, {hop}, {sigma}, {inner})")
    if case % 3 == 0:
        rng.normal(0, 1, (B, (F - 1) * hop + 1))
