"""Semi-supervised GAN discriminator / classifier of the reference's ``sgan.py`` (train step) on PyTorch-ROCm.

Architecture (sgan.py:132-217; images/sgan_d_model.png, sgan_c_model.png): per projection branch
3 x [Conv2D(128 / 64 / 32, 3x3, stride 2, 'same') + BatchNorm + LeakyReLU(0.2)] (128 -> 16); concatenate;
Flatten (NHWC, 16*16*96 = 24 576); 2 x [Dense 64 + BatchNorm + LeakyReLU(0.2) + Dropout 0.5]; Dense n_classes.
Two heads on the shared trunk: supervised ``c`` = softmax + sparse categorical cross-entropy; unsupervised
``d`` = custom_activation sum(exp)/(sum(exp)+1) (sgan.py:125-129) + binary cross-entropy.  Each head has its own
Adam(lr 2e-4, beta1 0.5) (sgan.py:206-207,214-215).  Keras semantics kept: TF 'same' padding, NHWC flatten,
RandomNormal(0, 0.02) kernels, zero biases, BatchNorm(momentum 0.99, eps 1e-3) = torch momentum 0.01,
Adam eps 1e-7.  BASELINE config 5: fp16 autocast + loss scaling, data parallel over the GPUs of a node with
DistributedDataParallel (backend "nccl" = RCCL all-reduce of ~1.86 M gradient elements per step);
BatchNorm statistics stay per replica (what Keras does per replica).
Only the discriminator/classifier train step is in scope; the generator is not.
"""
import os

import numpy as np

from .nn_common import make_same_conv, to_nchw, flatten_nhwc

RESCALE = (128, 128)        # sgan.py:39


def _nn():
    import torch.nn as nn
    return nn


class Discriminator(_nn().Module):
    def __init__(self, shapes=((128, 128, 1),) * 3, n_classes=3):
        nn = _nn()
        super().__init__()
        self.shapes = [tuple(s) for s in shapes]
        self.n_classes = n_classes
        self.branches = nn.ModuleList()
        feat = 0
        for (h, w, c) in self.shapes:
            layers, ch = [], c
            for out in (128, 64, 32):
                layers += [make_same_conv(ch, out, 3, 2), nn.BatchNorm2d(out, eps=1e-3, momentum=0.01), nn.LeakyReLU(0.2)]
                ch = out
                h, w = -(-h // 2), -(-w // 2)
            self.branches.append(nn.Sequential(*layers))
            feat += h * w * 32
        self.flat_features = feat
        self.fc1 = nn.Linear(feat, 64); self.bn1 = nn.BatchNorm1d(64, eps=1e-3, momentum=0.01)
        self.fc2 = nn.Linear(64, 64); self.bn2 = nn.BatchNorm1d(64, eps=1e-3, momentum=0.01)
        self.fc3 = nn.Linear(64, n_classes)
        self.act = nn.LeakyReLU(0.2)
        self.drop = nn.Dropout(0.5)
        # backward of the fused conv + batch-norm layers: zeros for the (cancelled) convolution bias, or no gradient at all
        # (nn_common.fused_step_scope; DiscriminatorTrainer turns the zeros off unless torch's DDP wraps the model)
        self.zero_bias_grads = True
        for mod in self.modules():
            if isinstance(mod, (nn.Conv2d, nn.Linear)):
                nn.init.normal_(mod.weight, 0.0, 0.02)      # RandomNormal(stddev=0.02), sgan.py:176
                nn.init.zeros_(mod.bias)

    @staticmethod
    def _branch(x, br, half=None):
        """[Conv2D 'same' s2 + BatchNorm + LeakyReLU] x 3 (sgan.py:137-158).  On the GPU under half-precision autocast and
        in training mode: batch norm + LeakyReLU + the bottom/right zero pad of the next convolution are one fused HIP op
        (nn_common.bn_lrelu_pad), the convolutions run without their bias (batch norm cancels it; its gradient is exactly
        zero and is handed back as such), and the 1-channel first layer is one autograd node whose backward sums the weight
        gradient without materialising the gradient of the convolution output (nn_common.conv1_bn_lrelu_pad).
        Otherwise the plain PyTorch layers run."""
        import torch
        import torch.nn.functional as F
        from .nn_common import bn_lrelu_pad, conv1_bn_lrelu_pad, tf_same_pad
        layers = list(br)
        nlayer = len(layers) // 3
        even = all(s % (2 ** nlayer) == 0 for s in x.shape[-2:])
        adt = torch.get_autocast_dtype("cuda") if (x.is_cuda and torch.is_autocast_enabled("cuda")) else None
        if not (x.is_cuda and br.training and even and adt in (torch.float16, torch.bfloat16)):
            return br(x)
        ph, pw = tf_same_pad(x.shape[-2], 3, 2), tf_same_pad(x.shape[-1], 3, 2)
        # the 1-channel input of the first convolution; tagged channels_last explicitly (for C = 1 the strides alone do
        # not say), otherwise MIOpen answers in NCHW and 268 MB layout copies appear on both sides of layer 1
        x = F.pad(x, (pw[0], pw[1], ph[0], ph[1])).contiguous(memory_format=torch.channels_last)
        for li in range(nlayer):
            conv, bn, act = layers[3 * li].conv, layers[3 * li + 1], layers[3 * li + 2]      # the inner Conv2d: input is padded
            pad = 1 if li + 1 < nlayer else 0
            c = conv.out_channels
            if (li == 0 and conv.in_channels == 1 and not x.requires_grad and x.shape[-1] % 2 == 1 and x.shape[-2] % 2 == 1
                    and c % 8 == 0 and 256 % (c // 8) == 0 and bn.track_running_stats and bn.affine and bn.momentum is not None):
                x = conv1_bn_lrelu_pad(x, conv, bn, act.negative_slope, pad, adt)
                continue
            z = F.conv2d(x, half.get(conv.weight, conv.weight) if half else conv.weight, None, stride=2)
            x = bn_lrelu_pad(z, bn, act.negative_slope, pad=pad, conv_bias=conv.bias)
        return x

    def forward(self, xz, yz, xy):
        """Pre-activation class scores (N, n_classes) -- the shared ``cls`` tensor of sgan.py:199."""
        import torch
        import torch.nn.functional as F
        from .nn_common import cast_all, fused_step_scope
        adt = torch.get_autocast_dtype("cuda") if (xz.is_cuda and torch.is_autocast_enabled("cuda")) else None
        if not (self.training and adt in (torch.float16, torch.bfloat16)):
            outs = [self._branch(x, br) for x, br in zip((xz, yz, xy), self.branches)]
            fv = flatten_nhwc(torch.cat(outs, dim=1))
            h = self.drop(self.act(self.bn1(self.fc1(fv))))
            h = self.drop(self.act(self.bn2(self.fc2(h))))
            return self.fc3(h)
        # training under half-precision autocast on the GPU: the weights the matrix cores read are cast in one launch (and
        # their gradients cast back in one), the batch-norm bookkeeping of the fused layers is applied in two (nn_common)
        ws = [layers[i].conv.weight for layers in (list(br) for br in self.branches) for i in range(3, len(layers), 3)]
        ws += [self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, self.fc3.weight, self.fc3.bias]
        half = dict(zip(ws, cast_all(adt, *ws)))
        with fused_step_scope(bias_grads=self.zero_bias_grads):
            outs = [self._branch(x, br, half) for x, br in zip((xz, yz, xy), self.branches)]
        fv = flatten_nhwc(torch.cat(outs, dim=1))
        h = self.drop(self.act(self.bn1(F.linear(fv, half[self.fc1.weight], half[self.fc1.bias]))))
        h = self.drop(self.act(self.bn2(F.linear(h, half[self.fc2.weight], half[self.fc2.bias]))))
        return F.linear(h, half[self.fc3.weight], half[self.fc3.bias])


    # ---- Keras layout in / out -------------------------------------------------------------------------
    def _conv_bn_pairs(self):
        out = []
        for br in self.branches:
            layers = list(br)
            out.append([(layers[i].conv, layers[i + 1]) for i in range(0, len(layers), 3)])
        return out

    def keras_weights(self):
        """The parameters AND BatchNorm statistics in Keras layout, float64 numpy:
        ``(branches, dense)`` with branches[b] = [(kernel (kh,kw,cin,cout), bias, gamma, beta, moving_mean, moving_variance)] x 3
        and dense = [(kernel (in,out), bias, gamma, beta, moving_mean, moving_variance)] x 2 + [(kernel, bias)] --
        what ``layer.get_weights()`` of the reference's c_model / d_model holds layer by layer (sgan.py:132-199;
        Keras BatchNormalization lists gamma, beta, moving_mean, moving_variance)."""
        def a(t):
            return t.detach().double().cpu().numpy()

        def bn4(bn):
            return a(bn.weight), a(bn.bias), a(bn.running_mean), a(bn.running_var)
        branches = [[(a(conv.weight.permute(2, 3, 1, 0)), a(conv.bias)) + bn4(bn) for conv, bn in pairs] for pairs in self._conv_bn_pairs()]
        dense = [(a(self.fc1.weight.t()), a(self.fc1.bias)) + bn4(self.bn1), (a(self.fc2.weight.t()), a(self.fc2.bias)) + bn4(self.bn2),
                 (a(self.fc3.weight.t()), a(self.fc3.bias))]
        return branches, dense

    def set_keras_weights(self, branches, dense):
        """Inverse of :meth:`keras_weights`: what a maintainer pulls out of a trained ``c_model_XXXX.h5`` (sgan.py:496-500)
        goes in here; the moving statistics land in the BatchNorm buffers, so inference (``predict`` / ``evaluate``)
        reproduces the Keras model."""
        import torch
        pairs = self._conv_bn_pairs()
        if len(branches) != len(pairs) or len(dense) != 3:
            raise ValueError("expected %d branches of 3 (conv + batch-norm) tuples and 3 dense tuples" % len(pairs))

        def put(dst, arr, what):
            t = torch.as_tensor(np.asarray(arr), dtype=dst.dtype)
            if tuple(t.shape) != tuple(dst.shape):
                raise ValueError("%s: Keras array gives %s, the layer holds %s" % (what, tuple(t.shape), tuple(dst.shape)))
            dst.copy_(t.to(dst.device))

        def put_bn(bn, g, b, mean, var, what):
            put(bn.weight, g, what + " gamma"); put(bn.bias, b, what + " beta")
            put(bn.running_mean, mean, what + " moving_mean"); put(bn.running_var, var, what + " moving_variance")

        with torch.no_grad():
            for bi, (prs, given) in enumerate(zip(pairs, branches)):
                if len(given) != len(prs):
                    raise ValueError("branch %d: expected %d (conv + batch-norm) tuples" % (bi, len(prs)))
                for li, ((conv, bn), (k, b, g, be, mu, var)) in enumerate(zip(prs, given)):
                    what = "branch %d layer %d" % (bi, li)
                    put(conv.weight, np.asarray(k).transpose(3, 2, 0, 1), what + " kernel")
                    put(conv.bias, b, what + " bias")
                    put_bn(bn, g, be, mu, var, what)
            for fc, bn, tup, nm in ((self.fc1, self.bn1, dense[0], "dense"), (self.fc2, self.bn2, dense[1], "dense_1")):
                k, b, g, be, mu, var = tup
                put(fc.weight, np.asarray(k).T, nm + " kernel"); put(fc.bias, b, nm + " bias")
                put_bn(bn, g, be, mu, var, nm)
            k, b = dense[2]
            put(self.fc3.weight, np.asarray(k).T, "dense_2 kernel"); put(self.fc3.bias, b, "dense_2 bias")
        return self


def class_weight_to_sample_weight(y, class_weight):
    """``train_on_batch(..., class_weight=w)`` as Keras applies it (sgan.py:529-530 passes the data set's class weights to
    the d update): the target is cast to an integer class, ``int(y)`` truncating toward zero -- the smoothed real labels in
    [0.7, 1.2) of sgan.py:396-398 therefore select class 0 or 1 -- and that class's weight becomes the sample's weight
    (classes missing from the dict weigh 1)."""
    yi = np.trunc(np.asarray(y, dtype=np.float64).reshape(-1)).astype(np.int64)
    w = np.ones(len(yi), dtype=np.float32)
    for cls, val in dict(class_weight).items():
        w[yi == int(cls)] = float(val)
    return w


def custom_activation(logits):
    """sgan.py:125-129: sum(exp)/(sum(exp)+1) = sigmoid(logsumexp(logits))."""
    import torch
    return torch.sigmoid(torch.logsumexp(logits.float(), dim=-1, keepdim=True))


def c_loss(logits, y):
    """sparse_categorical_crossentropy on the softmax head (sgan.py:205)."""
    import torch.nn.functional as F
    return F.cross_entropy(logits.float(), y)


def d_loss(logits, y, sample_weight=None):
    """binary_crossentropy on the custom-activation head (sgan.py:213), evaluated on the logit
    lse = logsumexp(logits): -y log D - (1-y) log(1-D) = softplus(lse) - y*lse.  Labels may be smoothed
    floats outside [0,1] (sgan.py:396-403)."""
    import torch
    import torch.nn.functional as F
    lse = torch.logsumexp(logits.float(), dim=-1)
    loss = F.softplus(lse) - y.float().reshape(-1) * lse
    if sample_weight is not None:
        loss = loss * sample_weight.float().reshape(-1)
    return loss.mean()


class DiscriminatorTrainer:
    """c_model / d_model ``train_on_batch`` (sgan.py:525-532) with the reference's optimizers, fp16 autocast and,
    when torch.distributed is initialised, DistributedDataParallel gradient all-reduce."""

    def __init__(self, model, lr=2e-4, beta1=0.5, amp_dtype="float16", ddp=None, use_graph=False, tune_convolutions=False):
        """``use_graph``: capture forward + backward of each head in a HIP graph (torch.cuda.graphs) after three eager
        warm-up steps and replay it afterwards; the optimizer, the loss scaler and the gradient all-reduce stay outside
        the graph.  Inputs must keep their shapes.  With the fused layers the step is launch-bound on the host side,
        which is what the graph removes.

        ``ddp``: data parallelism when torch.distributed is initialised with more than one rank (None = on).  The
        replicas are kept in step the MI355X way: every parameter's ``.grad`` is a view into ONE flat float32 bucket
        (1.86 M elements = 7.4 MB) that is all-reduced (RCCL over xGMI; gloo on CPU) once per update, after the
        backward pass and before the loss-scaled optimizer step -- a single collective of a few tens of microseconds
        instead of per-bucket hooks inside the backward, so forward + backward can still be replayed from a HIP graph.
        ``ddp="torch"`` wraps the model in torch's DistributedDataParallel instead (no graph replay then).
        BatchNorm statistics stay per replica (what Keras does per replica).

        ``tune_convolutions``: runs this trainer's steps with ``torch.backends.cudnn.benchmark = True`` (set around each step and
        restored afterwards: the flag is process-wide and changes which MIOpen kernels -- and so which round-off -- every other
        convolution in the process gets): MIOpen then times its solvers for every convolution shape on first use instead of taking
        its heuristic pick -- a few seconds once, 3 % off the step on an MI355X (a CK xdl forward kernel and a smaller-tile
        weight-gradient kernel win)."""
        import torch
        import torch.distributed as dist
        self._tune = bool(tune_convolutions)
        self.model = model
        self.device = next(model.parameters()).device
        self.net = model
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.world = dist.get_world_size() if multi else 1
        mode = ddp
        if mode is None or mode is True:
            mode = "flat" if multi else None
        elif mode is False:
            mode = None
        if mode is not None and not multi:
            mode = None
        self.ddp_mode = mode
        self._flat = None
        model.zero_bias_grads = mode == "torch"         # DDP wants a gradient for every parameter; Adam does not (exactly zero)
        if mode == "torch":
            from torch.nn.parallel import DistributedDataParallel as DDP
            self.net = DDP(model, device_ids=[self.device.index] if self.device.type == "cuda" else None,
                           gradient_as_bucket_view=True)
        elif mode == "flat":
            with torch.no_grad():                       # identical replicas to start from (what DDP's constructor does)
                for t in list(model.parameters()) + list(model.buffers()):
                    dist.broadcast(t.data, src=0)
            params = [p for p in model.parameters() if p.requires_grad]
            self._flat = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=self.device)
            off = 0
            for p in params:
                if p.dtype != torch.float32:
                    raise TypeError("flat gradient bucket: float32 master parameters expected")
                # the view takes the parameter's own strides (channels_last convolution kernels): the fused Adam and autograd's
                # gradient-layout contract want grad and param laid out alike
                p.grad = torch.as_strided(self._flat, p.size(), p.stride(), storage_offset=off)
                off += p.numel()
        elif mode is not None:
            raise ValueError("ddp must be None, False, 'flat' or 'torch'")
        # Adam(lr=0.0002, beta_1=0.5), Keras epsilon 1e-7 (sgan.py:206,214); one fused update kernel on the GPU
        # (the per-parameter kernels of the default implementation were 8 % of the step)
        fused = self.device.type == "cuda"
        self.opt_c = torch.optim.Adam(model.parameters(), lr=lr, betas=(beta1, 0.999), eps=1e-7, fused=fused)
        self.opt_d = torch.optim.Adam(model.parameters(), lr=lr, betas=(beta1, 0.999), eps=1e-7, fused=fused)
        self.amp_dtype = getattr(torch, amp_dtype) if (amp_dtype and self.device.type == "cuda") else None
        self.scaler = torch.amp.GradScaler("cuda", enabled=self.amp_dtype == torch.float16)
        self.use_graph = bool(use_graph) and self.device.type == "cuda" and mode != "torch"
        self._params = [q for q in model.parameters() if q.requires_grad]
        # on the GPU the update itself and the loss-scale rule run as three launches over all parameters (nn_common.DeviceAdam,
        # csrc/optim.hip); opt_c / opt_d stay the description of the optimizers (and the CPU path).  RML_DEVICE_ADAM=0: torch's.
        self._dev_adam = None
        if self.device.type == "cuda" and mode != "torch" and os.environ.get("RML_DEVICE_ADAM", "1") != "0":
            from .nn_common import DeviceAdam
            self._scale_t = torch.full((1,), 65536.0, dtype=torch.float32, device=self.device) if self.amp_dtype == torch.float16 else None
            shared = torch.zeros((3,), dtype=torch.int32, device=self.device)
            self._dev_adam = {id(o): DeviceAdam(self._params, lr, (beta1, 0.999), 1e-7, scale=self._scale_t, scaler_state=shared, mirror=o)
                              for o in (self.opt_c, self.opt_d)}
        self._graphs = {}           # head -> dict(graph, static inputs / targets, loss, logits, eager_calls)

    def _scaled(self, loss):
        if self._dev_adam is not None:
            return loss if self._scale_t is None else loss * self._scale_t
        return self.scaler.scale(loss)

    def _opt_step(self, opt):
        if self._dev_adam is not None:
            self._dev_adam[id(opt)].step()
            return
        self.scaler.step(opt)
        self.scaler.update()

    def _zero_grad(self, opt):
        if self._flat is not None:
            self._flat.zero_()          # the grads are views of the bucket: keep them, clear them in one kernel
        else:
            opt.zero_grad(set_to_none=not self.use_graph)

    def _allreduce_grads(self):
        """mean of the (loss-scaled) gradients over the replicas: one collective on the flat bucket.  An overflow on any
        rank reaches every rank through the sum, so the loss scaler skips the step everywhere."""
        if self._flat is None:
            return
        import torch.distributed as dist
        dist.all_reduce(self._flat, op=dist.ReduceOp.SUM)
        self._flat.mul_(1.0 / self.world)

    def _graph_step_impl(self, head, opt, make_loss, x, targets):
        """One update of ``head`` ('c' or 'd') through a captured graph.  ``targets``: tuple of tensors the loss needs
        (copied into static buffers); ``make_loss(logits, *static_targets)`` builds the loss."""
        import torch
        st = self._graphs.setdefault(head, {"eager": 0})
        xs = self._inputs(x)
        key = tuple(tuple(t.shape) for t in xs) + tuple(tuple(t.shape) for t in targets)
        if st.get("key") not in (None, key):            # shapes changed: start over
            st.clear(); st["eager"] = 0
        st["key"] = key
        self.net.train()
        if "graph" not in st:
            if st["eager"] < 3:                         # eager warm-up (MIOpen find, allocator, lazy initialisations)
                st["eager"] += 1
                self._zero_grad(opt)
                with torch.autocast("cuda", dtype=self.amp_dtype, enabled=self.amp_dtype is not None, cache_enabled=False):
                    logits = self.net(*xs)
                loss = make_loss(logits, *targets)
                self._scaled(loss).backward()
                self._allreduce_grads()
                self._opt_step(opt)
                return loss.detach(), logits.detach()
            st["xs"] = [t.clone() for t in xs]
            st["targets"] = [t.clone() for t in targets]
            if self._flat is None:
                # no gradient tensors during the capture: the backward pass then WRITES every gradient into a tensor of this
                # graph's pool instead of adding it to a zeroed one -- one small kernel per parameter (53 of them, 0.25 ms per
                # update in profiles/r03_stats_sgan.txt) and the zeroing pass less.  Each head keeps its own gradient tensors.
                for q in self._params:
                    q.grad = None
            else:
                self._zero_grad(opt)
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            # thread_local: the RCCL watchdog thread of a multi-rank job queries events while we capture; in the default
            # (global) mode that invalidates the capture
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                with torch.autocast("cuda", dtype=self.amp_dtype, enabled=self.amp_dtype is not None, cache_enabled=False):
                    logits = self.net(*st["xs"])
                loss = make_loss(logits, *st["targets"])
                self._scaled(loss).backward()
            st["graph"], st["loss"], st["logits"] = g, loss, logits
            st["grads"] = [q.grad for q in self._params] if self._flat is None else None
            # the capture itself does not run the kernels: fall through to the first replay
        for dst, src in zip(st["xs"], xs):
            dst.copy_(src)
        for dst, src in zip(st["targets"], targets):
            dst.copy_(src)
        if st["grads"] is None:
            self._zero_grad(opt)
        st["graph"].replay()
        if st["grads"] is not None:
            for q, gr in zip(self._params, st["grads"]):       # this head's gradients (the other head's graph owns other tensors)
                q.grad = gr
        self._allreduce_grads()
        self._opt_step(opt)
        return st["loss"].detach(), st["logits"].detach()

    def _tuned(self, fn, *args):
        """``fn(*args)`` with MIOpen's timed solver search on while this trainer runs its convolutions (``tune_convolutions``),
        the process-wide flag put back afterwards."""
        if not self._tune:
            return fn(*args)
        import torch
        prev = torch.backends.cudnn.benchmark
        torch.backends.cudnn.benchmark = True
        try:
            return fn(*args)
        finally:
            torch.backends.cudnn.benchmark = prev

    def _graph_step(self, head, opt, make_loss, x, targets):
        return self._tuned(self._graph_step_impl, head, opt, make_loss, x, targets)

    def _step(self, opt, loss_fn, x):
        return self._tuned(self._step_impl, opt, loss_fn, x)

    def _inputs(self, x):
        return [to_nchw(a, self.device) for a in x]

    def _step_impl(self, opt, loss_fn, x):
        import torch
        self.net.train()
        if self._flat is not None:
            self._zero_grad(opt)
        else:
            opt.zero_grad(set_to_none=True)
        xs = self._inputs(x)
        if self.amp_dtype is not None:
            with torch.autocast("cuda", dtype=self.amp_dtype):
                logits = self.net(*xs)
        else:
            logits = self.net(*xs)
        loss = loss_fn(logits)
        self._scaled(loss).backward()
        self._allreduce_grads()
        self._opt_step(opt)
        return loss.detach(), logits.detach()

    def train_on_batch_c(self, x, y, sync=True):
        """c_model.train_on_batch([xz,yz,xy], y) -> (loss, accuracy) as Python floats (Keras), or as 0-d CUDA tensors
        with ``sync=False`` (no host synchronisation: the next step's launches overlap this step's kernels)."""
        import torch
        yt = torch.as_tensor(np.asarray(y) if not isinstance(y, torch.Tensor) else y).to(self.device).long().reshape(-1)
        if self.use_graph:
            loss, logits = self._graph_step("c", self.opt_c, lambda lg, t: c_loss(lg, t), x, (yt,))
        else:
            loss, logits = self._step(self.opt_c, lambda lg: c_loss(lg, yt), x)
        acc = (logits.argmax(dim=-1) == yt).float().mean()
        return (float(loss), float(acc)) if sync else (loss, acc)

    def train_on_batch_d(self, x, y, sample_weight=None, sync=True, class_weight=None):
        """d_model.train_on_batch([xz,yz,xy], y[, class_weight=w_classes]) -> loss (float, or a 0-d CUDA tensor with
        ``sync=False``).  ``class_weight`` (sgan.py:529-530) becomes a per-sample weight the way Keras does it
        (:func:`class_weight_to_sample_weight`)."""
        import torch
        if class_weight is not None:
            if sample_weight is not None:
                raise ValueError("give class_weight or sample_weight, not both")
            sample_weight = class_weight_to_sample_weight(y.detach().cpu().numpy() if isinstance(y, torch.Tensor) else y, class_weight)
        yt = torch.as_tensor(np.asarray(y) if not isinstance(y, torch.Tensor) else y).to(self.device).float()
        sw = None if sample_weight is None else (sample_weight if isinstance(sample_weight, torch.Tensor)
                                                 else torch.as_tensor(np.asarray(sample_weight))).to(self.device)
        if self.use_graph:
            tg = (yt,) if sw is None else (yt, sw.float())
            loss, _ = self._graph_step("d" if sw is None else "dw", self.opt_d,
                                       (lambda lg, t: d_loss(lg, t)) if sw is None else (lambda lg, t, w_: d_loss(lg, t, w_)), x, tg)
        else:
            loss, _ = self._step(self.opt_d, lambda lg: d_loss(lg, yt, sw), x)
        return float(loss) if sync else loss

    def predict(self, x, batch_size=4096):
        """c_model.predict: softmax class probabilities, float32 numpy."""
        import torch
        self.model.eval()
        outs = []
        with torch.no_grad():
            for s in range(0, len(x[0]), batch_size):
                xs = self._inputs([a[s:s + batch_size] for a in x])
                if self.amp_dtype is not None:
                    with torch.autocast("cuda", dtype=self.amp_dtype):
                        lg = self.model(*xs)
                else:
                    lg = self.model(*xs)
                outs.append(torch.softmax(lg.float(), dim=-1).cpu())
        return torch.cat(outs).numpy()


def _evaluate_c(trainer, x, y, batch_size=4096):
    """c_model.evaluate([xz, yz, xy], y) (sgan.py:491: ``_, acc = c_model.evaluate(...)``): (sparse categorical
    cross-entropy, accuracy) over all samples, inference mode (BatchNorm moving statistics, no dropout)."""
    p = trainer.predict(x, batch_size=batch_size).astype(np.float64)
    yi = np.asarray(y).reshape(-1).astype(np.int64)
    if len(yi) != len(p):
        raise ValueError("evaluate: %d label(s) for %d sample(s)" % (len(yi), len(p)))
    if len(yi) == 0:
        return 0.0, 0.0
    pt = np.clip(p[np.arange(len(yi)), yi], 1e-7, 1.0 - 1e-7)
    return float(-np.log(pt).mean()), float((p.argmax(axis=1) == yi).mean())


DiscriminatorTrainer.evaluate = _evaluate_c


def define_discriminator(xz_shape=(128, 128, 1), yz_shape=(128, 128, 1), xy_shape=(128, 128, 1), n_classes=3, device=None):
    """Counterpart of sgan.define_discriminator (sgan.py:160): one shared trunk; the d / c heads are the two
    losses of :class:`DiscriminatorTrainer`."""
    import torch
    dev = torch.device(device) if device is not None else (
        torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu"))
    return Discriminator([xz_shape, yz_shape, xy_shape], n_classes).to(dev).to(memory_format=torch.channels_last)
