"""One-rank RCCL all-reduce on the library's weightUpdates arena aliased as a torch tensor (what bench.py
does with N ranks).  Run on a GPU box: python tools/nccl_one_rank.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, torch.distributed as dist
import __graft_entry__ as ge
from helpers import net_desc, random_sequences, random_weights
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
pkg = ge.load_package()
rng = np.random.RandomState(4)
layers = net_desc(5, [("blstm", 8)], 4); weights = random_weights(layers, rng, 0.3)
xs, ts = random_sequences(rng, [7, 5, 3], 5, C=4); frac = pkg.make_fraction(xs, ts, 3)
net = pkg.NeuralNetwork(layers, weights, 3, 7)
net.load_sequences(frac); net.compute_forward_pass(); net.compute_backward_pass()
w, g, d, n = net.param_arena()
grads = torch.as_tensor(pkg.parallel.DeviceArray(g, n), device="cuda")
before = pkg.parallel.flatten_updates([l.weight_updates() for l in net.trainable_layers()])
net.join()
with torch.cuda.stream(net.torch_stream(torch)):        # RCCL orders itself against the current torch stream
    dist.all_reduce(grads, op=dist.ReduceOp.SUM)
torch.cuda.synchronize()
after = grads.cpu().numpy()
assert np.array_equal(before, after), "1-rank all-reduce must be the identity"
net.update_weights_fused(1e-2, 0.9); net.synchronize()
print("nccl one-rank all-reduce on the aliased arena: ok,", n, "floats")
dist.destroy_process_group()
