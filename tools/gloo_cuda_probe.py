"""Does torch's gloo backend complete an asynchronous all-reduce / broadcast of GPU tensors when two ranks share one device?
(decides whether a rehearsal of the multi-rank bench on a one-GPU box can use gloo)"""
import faulthandler, os, sys, time
import torch, torch.distributed as dist
faulthandler.dump_traceback_later(60, exit=True)
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
torch.cuda.set_device(0)
big = torch.full((40 * 1024 * 1024,), float(rank + 1), device='cuda')
t = time.time(); dist.broadcast(big, src=0); torch.cuda.synchronize(); print(rank, 'broadcast %.2f s' % (time.time() - t), float(big[0]), flush=True)
hs = []
t = time.time()
for lo in range(0, big.numel(), 8 * 1024 * 1024):
    hs.append(dist.all_reduce(big[lo:lo + 8 * 1024 * 1024], async_op=True))
for h in hs: h.wait()
torch.cuda.synchronize(); print(rank, 'async all_reduce x%d %.2f s' % (len(hs), time.time() - t), float(big[0]), float(big[-1]), flush=True)
# the same behind about a second of queued torch kernels (no kernel of this repository involved)
a = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16)
torch.cuda.synchronize(); t = time.time()
for _ in range(400): a2 = a @ a
print(rank, 'queued in %.2f s' % (time.time() - t), flush=True)
hs = [dist.all_reduce(big[lo:lo + 8 * 1024 * 1024], async_op=True) for lo in range(0, big.numel(), 8 * 1024 * 1024)]
for h in hs: h.wait()
torch.cuda.synchronize(); print(rank, 'async all_reduce behind queued torch work %.2f s' % (time.time() - t), flush=True)
dist.destroy_process_group()
