import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
W = 64
send_host = torch.empty(W, dtype=torch.float64, pin_memory=True)
recv_host = torch.empty(W, dtype=torch.float64, pin_memory=True)
send = torch.empty(W, dtype=torch.float64, device=dev)
recv = torch.empty(W, dtype=torch.float64, device=dev)
class Mapped:
    def __init__(self, t):
        self.__cuda_array_interface__ = {"shape": tuple(t.shape), "typestr": "<f8", "data": (t.data_ptr(), False), "version": 2}
send_view = torch.as_tensor(Mapped(send_host), device=dev)
recv_view = torch.as_tensor(Mapped(recv_host), device=dev)
local = np.random.default_rng(0).normal(size=W)
def a():
    send_host.numpy()[:] = local
    send.copy_(send_host, non_blocking=True)
    dist.all_gather_into_tensor(recv, send)
    recv_host.copy_(recv, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return recv_host.numpy().tolist()
def b():
    send_host.numpy()[:] = local
    send.copy_(send_view)
    dist.all_gather_into_tensor(recv, send)
    recv_view.copy_(recv)
    torch.cuda.current_stream().synchronize()
    return recv_host.numpy().tolist()
def c():  # collective straight on the mapped views (NOT used: cannot be verified across ranks here)
    send_host.numpy()[:] = local
    dist.all_gather_into_tensor(recv_view, send_view)
    torch.cuda.current_stream().synchronize()
    return recv_host.numpy().tolist()
def d():  # only the collective + sync
    dist.all_gather_into_tensor(recv, send)
    torch.cuda.current_stream().synchronize()
for name, f in (("dma copies", a), ("kernel copies over mapped pinned memory", b), ("collective on mapped memory", c), ("collective only", d)):
    for _ in range(20): f()
    t0 = time.perf_counter()
    for _ in range(300): out = f()
    print(f"{name}: {(time.perf_counter() - t0) / 300 * 1e6:.1f} us", "ok" if out is None or np.allclose(out, local) else "WRONG")
dist.destroy_process_group()
