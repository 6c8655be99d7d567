"""How long one exchange of the K partial sums (5 K + 2 u64 words, K = 256: 10 KiB) takes as a one-shot mailbox exchange, with the
ranks inside ONE process on ONE GPU (what a one-GPU box can show: the kernel's own cost -- stores to N mailboxes, N flags, the
wait, N loads per word -- not xGMI).  usage: python3 tools/mailbox_latency.py [rounds]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_mailbox import Ranks

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 500
for world in (1, 2, 3):
    for words in (5 * 256 + 2, 5 * 2048 + 2):
        R = Ranks(world, timeout_ms=5000)
        bufs = [torch.zeros(words, dtype=torch.int64, device=R.dev) for _ in range(world)]
        R.all_reduce(bufs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(rounds):
            for r in range(world):
                R.L.cniic_comm_all_reduce(R.comms[r], C.c_void_p(bufs[r].data_ptr()), C.c_uint64(words), C.c_int32(8))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("ranks %d  %6d words  %.2f us per exchange (all ranks, enqueue included)" % (world, words, dt / rounds * 1e6), flush=True)
        R.close()
