"""Which rounding of the bf16 CNN chain makes its probability error (session r5l)?  The float64 layers of Classifier.forward_exact
with a bf16 rounding inserted at ONE place at a time (inputs, conv1 weights, conv1 activations, conv2 weights, features, fc1
weights), on a trained model, against the pure float64 result; then all of them (= what the HIP chain does) and fp16 instead."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
import radar_ml_amd as rml
dnn = importlib.import_module("radar_ml_amd.dnn")
nnc = importlib.import_module("radar_ml_amd.nn_common")
dev = torch.device("cuda:0")
torch.manual_seed(1234)
m = dnn.define_classifier(device=dev)
X, Y, Z = 22, 31, 176
tv, tcls = rml.synth_volumes(1024, X, Y, Z, seed=1239, frame0=1 << 41, device=dev)
xs = [t.reshape(-1, 1, 80, 80) for t in nnc.preprocess_features(rml.process_volumes(tv, mode="max", scale=False), (X, Y, Z), (80, 80), out_dtype="float32")]
ty = tcls.long()
opt = torch.optim.Adam(m.parameters(), lr=1e-3, betas=(0.5, 0.999), eps=1e-7)
g = torch.Generator(device=dev).manual_seed(1234)
m.train()
for _ in range(400):
    idx = torch.randint(0, 1024, (64,), device=dev, generator=g)
    opt.zero_grad(set_to_none=True)
    F.cross_entropy(m.logits(*[t[idx] for t in xs]).float(), ty[idx]).backward()
    opt.step()
m.eval()
V, _ = rml.synth_volumes(2048, X, Y, Z, seed=1241, device=dev)
planes = nnc.preprocess_features(rml.process_volumes(V, mode="max", scale=False), (X, Y, Z), (80, 80), out_dtype="float32")


def rnd(t, kind):
    if kind is None:
        return t
    return t.to(kind).to(torch.float64)


def forward(where, kind):
    """float64 layers; ``where``: set of stage names rounded to ``kind``"""
    w = m._exact_weights(torch.float64)
    outs = []
    for x, convs in zip(planes, w["conv"]):
        x = x.reshape(x.shape[0], 1, 80, 80).double()
        if "input" in where:
            x = rnd(x, kind)
        for li, (k2d, b, kh, kw) in enumerate(convs):
            if ("w1" in where and li == 0) or ("w2" in where and li == 1):
                k2d = rnd(k2d, kind)
            oh, ow = x.shape[-2] // 2, x.shape[-1] // 2
            xp = F.pad(x, (0, 1, 0, 1))
            cols = torch.stack([xp[:, :, ky:ky + 2 * oh - 1:2, kx:kx + 2 * ow - 1:2] for ky in range(kh) for kx in range(kw)], dim=2).reshape(x.shape[0], -1, oh * ow)
            x = F.relu(torch.matmul(k2d, cols) + b[None, :, None]).reshape(x.shape[0], k2d.shape[0], oh, ow)
            if ("a1" in where and li == 0) or ("feat" in where and li == 1):
                x = rnd(x, kind)
        outs.append(x)
    h = x.new_empty(0)
    h = torch.cat(outs, dim=1).permute(0, 2, 3, 1).reshape(outs[0].shape[0], -1)
    (w1, b1), (w2, b2), (w3, b3) = w["fc"]
    if "wfc1" in where:
        w1 = rnd(w1, kind)
    h = F.relu(F.linear(h, w1, b1)); h = F.relu(F.linear(h, w2, b2))
    return torch.softmax(F.linear(h, w3, b3), dim=-1)


with torch.no_grad():
    ref = forward(set(), None)
    t2 = ref.topk(2, dim=1).values
    print("trained model: accuracy n/a, mean top-2 margin %.3f, max logit-free check ok" % float((t2[:, 0] - t2[:, 1]).mean()))
    for kind, name in ((torch.bfloat16, "bf16"), (torch.float16, "fp16")):
        for st in ("input", "w1", "a1", "w2", "feat", "wfc1"):
            e = float((forward({st}, kind) - ref).abs().max())
            print("%s rounding at %-6s only: max |dp| = %.2e" % (name, st, e))
        e = float((forward({"input", "w1", "a1", "w2", "feat", "wfc1"}, kind) - ref).abs().max())
        print("%s rounding everywhere      : max |dp| = %.2e" % (name, e))
    hip = m.predict_volumes(V, label_guard=None).double()
    print("the HIP bf16 chain             : max |dp| = %.2e" % float((hip - ref).abs().max()))
    e = float((forward({"input", "w1", "a1", "w2", "wfc1"}, torch.bfloat16) - ref).abs().max())
    print("bf16 everywhere but the features: max |dp| = %.2e" % e)
