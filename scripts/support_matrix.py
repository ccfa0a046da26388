#!/usr/bin/env python3
"""Which (agent x basis family x weight mode x storage) combinations rsrl_hip_create accepts -- asked of the library itself, one small ctx each.
    python scripts/support_matrix.py > profiles/r05_support_matrix.md        (GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rsrl_amd as ra  # noqa: E402

AGENTS = [(0, "QLearning"), (1, "SARSA"), (2, "ExpectedSARSA"), (5, "PAL"), (3, "SARSALambda"), (4, "QLambda"), (6, "GreedyGQ"), (9, "QSigma"), (7, "TD"), (8, "TDLambda")]
FAMILIES = [
    ("Fourier, register family (MountainCar 1-5, CartPole / Acrobot 1)", dict(domain=0, order=5)),
    ("Fourier, generic orders (MountainCar 6-7, CartPole / Acrobot 2-6)", dict(domain=1, order=3)),
    ("Fourier order 7 on CartPole / Acrobot (wave family), f32", dict(domain=2, order=7)),
    ("... bf16 weights + stochastic rounding", dict(domain=2, order=7, weight_dtype=ra.W_BF16)),
    ("tile coding, per-learner tables", dict(domain=1, basis=ra.TILE_CODING)),
    ("ONE shared approximator, dense Fourier (register family)", dict(domain=0, order=5, weight_mode=ra.W_SHARED)),
    ("ONE shared approximator, tile coding", dict(domain=1, basis=ra.TILE_CODING, weight_mode=ra.W_SHARED)),
]


def main():
    print("# What `rsrl_hip_create` accepts (asked of the library, `scripts/support_matrix.py`; every accepted cell is covered by the campaigns of DESIGN §2)\n")
    print("| basis family / weights | " + " | ".join(n for _, n in AGENTS) + " |")
    print("|---|" + "---|" * len(AGENTS))
    notes = {}
    for fname, fkw in FAMILIES:
        row = []
        for algo, aname in AGENTS:
            kw = dict(fkw, algo=algo, n_envs=8, policy=ra.RANDOM if algo in (7, 8) else 1, lam=0.5, lr_td=0.001, n_steps=2, sigma=0.5)
            try:
                with ra.Context(**kw):
                    row.append("yes")
            except ra.RsrlHipError as e:
                key = str(e).split(": ", 1)[-1][:160]
                idx = notes.setdefault(key, len(notes) + 1)
                row.append(f"no [{idx}]")
        print(f"| {fname} | " + " | ".join(row) + " |")
    print()
    for key, idx in notes.items():
        print(f"[{idx}] {key}")
    print("\nThe per-learner epsilon schedule (`epsilon_decay`): one-step agents and SARSALambda / QLambda on the register family, one-step agents on tile coding and the generic orders.")


if __name__ == "__main__":
    main()
