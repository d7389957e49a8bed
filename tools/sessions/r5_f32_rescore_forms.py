"""Forms of the guard's float32 layers (session r5p): the three branches as separate convolutions (the module's own layers), as ONE
grouped convolution per layer (groups = 3), channels_last or not; time per 256 / 576 rows and the difference to float64."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
import radar_ml_amd as rml
dnn = importlib.import_module("radar_ml_amd.dnn")
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = dnn.define_classifier(device=dev).eval()


def T(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r


w1 = torch.cat([br[0].conv.weight.detach() for br in m.branches]).float().contiguous()      # (192, 1, 3, 3)
b1 = torch.cat([br[0].conv.bias.detach() for br in m.branches]).float()
w2 = torch.cat([br[1].conv.weight.detach() for br in m.branches]).float().contiguous()      # (96, 64, 3, 3)
b2 = torch.cat([br[1].conv.bias.detach() for br in m.branches]).float()


def grouped(xs, cl):
    x = torch.stack([t.reshape(t.shape[0], 80, 80) for t in xs], dim=1).float()              # (N, 3, 80, 80)
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
    x = F.relu(F.conv2d(F.pad(x, (0, 1, 0, 1)), w1, b1, stride=2, groups=3))
    x = F.relu(F.conv2d(F.pad(x, (0, 1, 0, 1)), w2, b2, stride=2, groups=3))
    h = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)
    h = F.relu(F.linear(h, m.fc1.weight, m.fc1.bias)); h = F.relu(F.linear(h, m.fc2.weight, m.fc2.bias))
    return torch.softmax(F.linear(h, m.fc3.weight, m.fc3.bias), dim=-1)


with torch.no_grad():
    for k in (256, 576):
        xs = [torch.rand(k, 80, 80, device=dev) * 2 - 1 for _ in range(3)]
        ref = m.forward_float64(*xs)
        ms, p = T(lambda: m.forward_exact(*xs, precision="float32")); print("k=%d module layers       %.3f ms  |d64| %.1e" % (k, ms, float((p.double() - ref).abs().max())))
        for cl in (False, True):
            ms, p = T(lambda: grouped(xs, cl)); print("k=%d grouped conv (cl=%d)   %.3f ms  |d64| %.1e" % (k, int(cl), ms, float((p.double() - ref).abs().max())))
        torch.backends.cudnn.benchmark = True
        ms, p = T(lambda: m.forward_exact(*xs, precision="float32")); print("k=%d module layers, benchmark mode %.3f ms" % (k, ms))
        ms, p = T(lambda: grouped(xs, True)); print("k=%d grouped cl, benchmark mode    %.3f ms  |d64| %.1e" % (k, ms, float((p.double() - ref).abs().max())))
        torch.backends.cudnn.benchmark = False
