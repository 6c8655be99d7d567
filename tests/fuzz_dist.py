"""Multi-rank fuzz on one GPU: the library's K-means loop over the host-transport communicator or the one-shot mailbox exchange (2-3 ranks), random images and
K, against the oracle's clustering of the union.  usage: python tests/fuzz_dist.py [cases] [seed]   (test infrastructure: the oracle is the checker)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))   # (this directory)


def main():
    import numpy as np
    import test_dist as T
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
    bad = 0
    for i in range(cases):
        world = int(rng.choice([2, 3]))
        K = int(rng.choice([2, 5, 16, 40]))
        h, w = int(rng.integers(8, 90)), int(rng.integers(8, 90))
        env = {"TEST_COLLECTIVES": str(rng.choice(["host", "mailbox"])), "CNIIC_COLLECTIVE_TIMEOUT_MS": "20000", "CNIIC_SP_MIN_PIXELS": str(int(rng.choice([0, 1 << 40]))),
               "FUZZ_SEED0": str(int(rng.integers(0, 1000))), "FUZZ_H": str(h), "FUZZ_W": str(w)}
        os.environ.update({k: v for k, v in env.items() if k.startswith("FUZZ_")})   # make_img here = make_img in the workers
        res = T._run(world, K, use_hip=True, env=env)
        exp, iters = T.expected_streams([T.make_img(r) for r in range(world)], K)
        ok = all(res[r][0] == exp[r] and res[r][1] == iters for r in range(world))
        if not ok:
            bad += 1
            print(json.dumps(dict(case=i, world=world, K=K, env=env)))
    print(json.dumps(dict(cases=cases, mismatches=bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
