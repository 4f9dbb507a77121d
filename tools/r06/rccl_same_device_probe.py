"""Round 6: can two RCCL ranks share ONE device — so that cov_gather's RCCL branch with n > 1 could be driven on the
one-GPU lease box?  Two processes, both on cuda:0, backend nccl (= RCCL), one all_reduce; and ncclCommInitAll over the device list {0, 0}
through libcovermhip's own dlopen'ed RCCL (COVERM_FORCE_RCCL, two sessions on device 0).  Prints what happened as JSON.

    python tools/r06/rccl_same_device_probe.py
"""
import json
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = {"rank": rank}
    try:
        import datetime
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
        t = torch.ones(4, device="cuda:0")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        res["all_reduce"] = t.tolist()
        dist.destroy_process_group()
    except Exception as ex:
        res["error"] = (type(ex).__name__ + ": " + str(ex))[:600]
    q.put(res)


def main():
    import torch.multiprocessing as mp
    out = {}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, 2, 29731, q)) for r in range(2)]
    [p.start() for p in ps]
    got = []
    for p in ps:
        p.join(120)
        if p.is_alive():
            p.kill()
    while not q.empty():
        got.append(q.get())
    out["torch_distributed_nccl_two_ranks_on_cuda0"] = got or "no result (ranks hung until the 120 s join; killed)"
    # libcovermhip: cov_gather over two sessions of device 0 with COVERM_FORCE_RCCL — the code takes the RCCL branch only for DISTINCT devices
    try:
        import ctypes as C
        import numpy as np
        os.environ["COVERM_FORCE_RCCL"] = "1"
        from coverm_amd import synth
        from coverm_amd.engine import FilterConfig, Session
        ref = synth.make_reference(20, 1_000_000, seed=3, min_len=1500, max_len=200_000)
        b = synth.make_reads(ref, 20_000, seed=4)
        ss = []
        for k in range(2):
            s = Session(0, FilterConfig(), 75, want_hist=False)
            s.set_targets(ref.lengths); s.push(b); s.finish(); ss.append(s)
        L = ss[0]._lib
        arr = (C.c_void_p * 2)(ss[0]._h, ss[1]._h)
        L.cov_gather.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        rc = L.cov_gather(arr, 2, 0)
        L.cov_last_error.restype = C.c_char_p
        out["cov_gather_two_sessions_of_device_0"] = {"rc": int(rc), "last_error": (L.cov_last_error(ss[0]._h) or b"").decode()[:300],
                                                        "note": "cov_gather uses RCCL only when the sessions' devices are distinct (ncclCommInitAll refuses a device list with duplicates); the same-device sessions meet through device-to-device copies"}
        for s in ss:
            s.close()
    except Exception:
        out["cov_gather_two_sessions_of_device_0"] = {"error": traceback.format_exc()[-800:]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
