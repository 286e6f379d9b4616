"""Host-side driver of one UPP training step (the per-step recipe of reference
tools/runner_module.py:193-212) built for MI355X:

  * PEFT freezing by substring match BEFORE the gradient exchange is set up
    (reference :62-73 freezes after wrapping in DDP, which would all-reduce all 30.4 M params);
  * every trainable .grad is a view into ONE flat buffer -> one RCCL all-reduce per step;
  * all shapes of the step are functions of the config only, so forward + loss + backward are
    captured once into a HIP graph and replayed: ~2,700 kernel launches per step become one
    graph launch (the eager step is launch-bound: 32 ms wall vs 24 ms of kernels).  The
    all-reduce stays outside the graphs; clip + AdamW are a second graph.
"""
import os
import torch
import torch.distributed as dist

from utils.dist_utils import FlatGradAllReduce, flat_offsets

from . import functional as HF

PEFT_STAGE1 = ['downstream_adapter', 'downstream_adapter1', 'downstream_prompts', 'bnorm', 'cls_pos', 'cls_token',
               'cls_head_finetune']  # reference tools/runner_module.py:62-66


def freeze_for_peft(model, keys=PEFT_STAGE1):
    n = 0
    for name, p in model.named_parameters():
        on = any(k in name for k in keys)
        p.requires_grad_(on)
        n += p.numel() if on else 0
    return n


def make_adamw(model, lr=5e-4, weight_decay=0.05, capturable=False):
    """Two groups (reference tools/builder.py:40-55): no decay for 1-D / bias / 'token' parameters."""
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (no_decay if (p.dim() == 1 or name.endswith(".bias") or 'token' in name) else decay).append(p)
    groups = [{'params': no_decay, 'weight_decay': 0.}, {'params': decay, 'weight_decay': weight_decay}]
    return torch.optim.AdamW(groups, lr=lr, capturable=capturable)


def split_decay(model):
    """(no_decay, decay) trainable parameters by the reference's rule (tools/builder.py:44-49)."""
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (no_decay if (p.dim() == 1 or name.endswith(".bias") or 'token' in name) else decay).append(p)
    return no_decay, decay


def _copy_many(dsts, srcs):
    """dst_j <- src_j; the contiguous HIP pairs in launches of 64 (upp_copy_batched), the rest one by one."""
    batched = [(d, s) for d, s in zip(dsts, srcs)
               if d.is_cuda and s.is_cuda and d.dtype == s.dtype and d.shape == s.shape and d.is_contiguous() and s.is_contiguous() and d.numel() > 0]
    taken = {id(d) for d, _ in batched}
    if batched:
        HF.ops.copy_batched([d for d, _ in batched], [s.detach() for _, s in batched])
    for d, s in zip(dsts, srcs):
        if id(d) not in taken:
            d.copy_(s)


class FlatAdamW:
    """clip_grad_norm_ + AdamW as three gfx950 kernels (csrc/optim.hip) over flat buffers: the parameters are
    re-pointed into one flat buffer (no-decay group first), the gradients are FlatGradAllReduce's buffer."""

    def __init__(self, no_decay, decay, flat_grad, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05, max_norm=10.0):
        from . import ops, _abi
        self._ops = ops
        params = list(no_decay) + list(decay)
        dev = params[0].device
        # same layout as the gradient buffer (FlatGradAllReduce over the same list): 16-byte aligned starts, zero gaps
        self._offsets, self.n = flat_offsets(params)
        self.split = self._offsets[len(no_decay)] if len(decay) else self.n
        self.p = torch.zeros(self.n, device=dev)
        with torch.no_grad():
            views = [self.p[off:off + q.numel()].view_as(q) for q, off in zip(params, self._offsets)]
            _copy_many(views, [q.detach() for q in params])            # (one launch per 64 parameters, not one runtime copy each)
            for q, view in zip(params, views):
                q.data = view                      # the model now reads its trainable weights from the flat buffer
        self._shapes = [tuple(q.shape) for q in params]
        self._n_no_decay = len(no_decay)
        self.g = flat_grad
        self.m = torch.zeros(self.n, device=dev)
        self.v = torch.zeros(self.n, device=dev)
        self.state = torch.zeros(8, device=dev)        # [0] step ... [5] learning rate, [6] weight decay (read by the kernel)
        self.scratch = torch.empty(int(_abi.load().upp_adamw_scratch_floats()), device=dev)
        self.hyper = (lr, betas[0], betas[1], eps, weight_decay, -1.0 if max_norm is None else max_norm)
        self._push_hyper()

    def _push_hyper(self):
        """lr and weight decay live in the device state buffer (upp_adamw_flat with lr < 0 reads them there), so a step
        captured in a HIP graph follows set_lr() / load_state_dict() / a scheduler without being re-captured."""
        self.state[5:7].copy_(torch.tensor([self.hyper[0], self.hyper[4]], dtype=torch.float32), non_blocking=False)

    def set_lr(self, lr, weight_decay=None):
        h = list(self.hyper)
        h[0] = float(lr)
        if weight_decay is not None:
            h[4] = float(weight_decay)
        self.hyper = tuple(h)
        self._push_hyper()
        self._sync_groups()

    def _sync_groups(self):
        """Keep the torch.optim-style view in step with `hyper` (set_lr, load_state_dict): a later sync_param_groups() must not push
        stale values -- e.g. the weight decay a checkpoint restored -- back to the device."""
        if hasattr(self, '_groups'):
            self._groups[0].update(lr=self.hyper[0], weight_decay=0.0)
            self._groups[1].update(lr=self.hyper[0], weight_decay=self.hyper[4])

    @property
    def param_groups(self):
        """torch.optim-style view for schedulers that do `for g in opt.param_groups: g['lr'] = ...`; call sync_param_groups()
        (or set_lr) afterwards to publish the values to the device."""
        if not hasattr(self, '_groups'):
            self._groups = [{'lr': self.hyper[0], 'weight_decay': 0.0}, {'lr': self.hyper[0], 'weight_decay': self.hyper[4]}]
        return self._groups

    def sync_param_groups(self):
        g = self.param_groups
        self.set_lr(g[-1]['lr'], g[-1]['weight_decay'])

    def step(self):
        _, b1, b2, eps, _, mn = self.hyper
        self._ops.adamw_flat(self.p, self.g, self.m, self.v, self.n, self.split, self.state, self.scratch, -1.0, b1, b2, eps, 0.0, mn)

    # -- checkpoint contract: the state_dict layout of torch.optim.AdamW with the two groups of make_adamw, so a
    #    checkpoint written by the reference loop (tools/builder.py:131-140) resumes here and vice versa
    def state_dict(self):
        lr, b1, b2, eps, wd, _ = self.hyper
        step = self.state[0].detach().cpu().clone()
        state = {}
        for i, (shp, off) in enumerate(zip(self._shapes, self._offsets)):
            n = int(torch.tensor(shp).prod()) if shp else 1
            state[i] = {'step': step.clone(), 'exp_avg': self.m[off:off + n].view(shp).clone(),
                        'exp_avg_sq': self.v[off:off + n].view(shp).clone()}
        k = self._n_no_decay
        common = dict(lr=lr, betas=(b1, b2), eps=eps, amsgrad=False, maximize=False, foreach=None, capturable=False,
                      differentiable=False, fused=None)
        groups = [dict(common, weight_decay=0., params=list(range(k))),
                  dict(common, weight_decay=wd, params=list(range(k, len(self._shapes))))]
        return {'state': state, 'param_groups': groups}

    def load_state_dict(self, sd):
        groups = sd['param_groups']
        order = [i for g in groups for i in g['params']]
        if len(order) != len(self._shapes):
            raise ValueError("optimizer state has %d parameters, expected %d" % (len(order), len(self._shapes)))
        step = 0.0
        with torch.no_grad():
            for i, shp, off in zip(order, self._shapes, self._offsets):
                n = int(torch.tensor(shp).prod()) if shp else 1
                st = sd['state'].get(i)
                if st is not None:
                    if tuple(st['exp_avg'].shape) != shp:
                        raise ValueError("optimizer state %d has shape %s, expected %s" % (i, tuple(st['exp_avg'].shape), shp))
                    self.m[off:off + n].copy_(st['exp_avg'].reshape(-1))
                    self.v[off:off + n].copy_(st['exp_avg_sq'].reshape(-1))
                    step = max(step, float(st['step']))
            self.state[0] = step
        g0 = groups[0]
        if (g0['betas'][0], g0['betas'][1], g0['eps']) != self.hyper[1:4] and getattr(self, 'captured', False):
            raise RuntimeError("betas / eps differ from the captured optimizer step: build the TrainStep after loading the checkpoint")
        self.hyper = (g0['lr'], g0['betas'][0], g0['betas'][1], g0['eps'], groups[-1]['weight_decay'], self.hyper[5])
        self._push_hyper()                              # lr / weight decay reach captured graphs through the device buffer
        self._sync_groups()


class _TrainingState:
    """Everything a training step mutates -- flat parameters, Adam moments and counters (or the library optimizer's state), every
    module buffer (BatchNorm statistics and counters), parameters outside the flat buffer -- cloned, to be put back IN PLACE.
    The graph-capture warm-up runs real steps (it has to: it sizes the uniform bank, tunes the remaining library GEMMs, fills
    the transposed-weight cache); with this it leaves no trace: the first step() starts from exactly the state the caller
    built -- on every rank, whatever data the ranks hold (a warm-up that applied two un-reduced updates made replicas diverge)."""

    def __init__(self, model, opt, flat):
        self.tensors = [b for b in model.buffers()] + [flat]
        if isinstance(opt, FlatAdamW):
            self.tensors += [opt.p, opt.m, opt.v, opt.state]
            self.opt, self.opt_state = None, None
        else:
            import copy
            self.tensors += [p for p in model.parameters() if p.requires_grad]
            self.opt, self.opt_state = opt, copy.deepcopy(opt.state_dict())
        with torch.no_grad():
            self.saved = [torch.empty_like(t, memory_format=torch.contiguous_format) for t in self.tensors]
            self._copy(self.saved, self.tensors)

    @staticmethod
    def _copy(dsts, srcs):
        """(one runtime copy per tensor is ~550 launches per snapshot of the headline model, which a kernel trace of a short run then shows
        as its most frequent 'kernel')"""
        _copy_many(dsts, srcs)

    def restore(self):
        with torch.no_grad():
            self._copy(self.tensors, self.saved)
        if self.opt is not None:
            self.opt.load_state_dict(self.opt_state)


def broadcast_model(model, src=0):
    """Rank `src`'s parameters and buffers to every rank (what DistributedDataParallel does when it wraps a model,
    reference tools/runner_module.py:53-57): replicas start identical whatever seed / checkpoint each process used."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    if dist.get_world_size() == 1 and os.environ.get("UPP_FORCE_DIST") != "1":        # (the one-rank RCCL rehearsal runs the broadcasts)
        return
    with torch.no_grad():
        for t in list(model.parameters()) + list(model.buffers()):
            if t.is_floating_point() or t.dtype in (torch.int64, torch.int32):
                dist.broadcast(t.data, src)


class TrainStep:
    """step(pts, labels) -> loss (device tensor).  With use_graph=True the inputs are copied into
    static buffers and the captured graphs are replayed.

    Other recipes than the classification one (pre-task Chamfer training, Point-MAE pre-training, part segmentation)
    pass `loss_fn(model, *inputs) -> (loss, metric)` and `inputs` (example tensors fixing shapes and dtypes); they are
    driven with step_inputs(*tensors)."""

    def __init__(self, model, batch_shape, grad_clip=10.0, use_graph=True, forward_kwargs=None, lr=5e-4, loss_fn=None,
                 inputs=None):
        self.model = model
        self.device = next(model.parameters()).device
        self.grad_clip = grad_clip
        self.kw = forward_kwargs or dict(completion_prompt=True, denoise=True, point_num=1024)
        self.use_graph = bool(use_graph) and self.device.type == 'cuda'
        no_decay, decay = split_decay(model)
        self.flat = FlatGradAllReduce(no_decay + decay)          # flat order: no-decay group, then decay group
        self.trainable = self.flat.params
        if self.device.type == 'cuda':
            self.opt = FlatAdamW(no_decay, decay, self.flat.flat, lr=lr, max_norm=grad_clip)
        else:                                                   # host runs (gloo tests, CPU baseline): library optimizer
            self.opt = make_adamw(model, lr=lr)
        # (UPP_FORCE_DIST=1: a one-rank group takes the N > 1 path too -- the RCCL rehearsal of a one-GPU box, tests/test_gpu_dist.py)
        self.distributed = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("UPP_FORCE_DIST") == "1")
        if self.distributed:
            broadcast_model(model)      # (after FlatAdamW re-pointed the trainable parameters into its flat buffer: views of it)
        self.pts = torch.zeros(batch_shape, device=self.device)
        self.labels = torch.zeros(batch_shape[0], dtype=torch.long, device=self.device)
        self.loss = torch.zeros((), device=self.device)
        self.loss_fn = loss_fn
        self.inputs = [t.detach().clone().to(self.device) for t in (inputs or [])]
        self._g_fb = self._g_opt = None
        self._grad_targets = ({p.data_ptr(): v for p, v in zip(self.trainable, self.flat.views)}
                              if self.device.type == 'cuda' else None)
        if self.device.type == 'cuda':
            # derived copies of frozen weights (W^T, bf16 plane images) that an earlier eager forward cached are brought up to date before
            # anything is captured: a weight written through `.data` since then moved no version counter (ops._WeightPlanes CONTRACT)
            HF.refresh_caches(model)

    # -- the two halves of a step ------------------------------------------------------------
    _seed = None

    def _forward_backward(self, pts=None, labels=None, kw=None):
        pts = self.pts if pts is None else pts
        labels = self.labels if labels is None else labels
        self.flat.zero()
        for p in self.trainable:
            p.grad = None              # let autograd write fresh gradients: no per-parameter accumulate kernels
        # W^T copies of the trainable weights (data-gradient GEMMs): persistent inside this driver, all refreshed by one launch here,
        # i.e. after whatever changed the weights since the last step (optimizer, load_state_dict, a restore)
        was, was_p = HF.TRANSPOSED.managed, HF.ops.PLANES.managed
        HF.TRANSPOSED.managed = HF.ops.PLANES.managed = True
        try:
            HF.TRANSPOSED.refresh_trainable()
            HF.ops.PLANES.refresh_trainable()          # (the bf16 plane images of the trainable weights and of their transposes: one launch)
            if self.loss_fn is not None:
                loss, acc = self.loss_fn(self.model, *self.inputs)
            else:
                logits = self.model.forward_tokens(*pts) if isinstance(pts, tuple) else self.model(pts, **(self.kw if kw is None else kw))
                loss, acc = self.model.get_loss_acc(logits, labels)
            # partial-sum reductions of parameter gradients: one launch after the pass, straight into the (zeroed) flat buffer
            with HF.deferred_sums(self._grad_targets) as scope:
                # (loss.backward() builds its seed with a fill launch; a persistent one is handed over instead)
                if self._seed is None or self._seed.device != loss.device or self._seed.dtype != loss.dtype:
                    self._seed = torch.ones((), dtype=loss.dtype, device=loss.device)
                torch.autograd.backward(loss, grad_tensors=self._seed)
        finally:
            HF.TRANSPOSED.managed, HF.ops.PLANES.managed = was, was_p
        # by-products the model keeps of its last forward (`aux`: the completion prompter's rebuilt points) must not keep this pass's autograd
        # graph alive: its AccumulateGrad nodes would be reused by the next pass with the stream THIS pass ran on -- under capture that is a
        # fork into the warm-up's stream inside the graph (torch warns "AccumulateGrad node's stream does not match"; a fork costs 65-115 us)
        aux = getattr(self.model, "aux", None)
        if isinstance(aux, dict):
            for k_, v_ in list(aux.items()):
                if isinstance(v_, torch.Tensor) and v_.grad_fn is not None:
                    aux[k_] = v_.detach()
        got = [(v, p.grad) for p, v in zip(self.trainable, self.flat.views) if p.grad is not None and p.data_ptr() not in scope.routed]
        # into the flat buffer in launches of 64 (upp_copy_batched): torch._foreach_copy_ issues one runtime copy per tensor here -- 55 ...
        # 100 `__amd_rocclr_copyBuffer` launches per step, 0.2 ... 0.35 ms in the round-2 / early round-3 kernel summaries
        _TrainingState._copy([v for v, _ in got], [g for _, g in got])
        both = [(v, p.grad) for p, v in zip(self.trainable, self.flat.views) if p.grad is not None and p.data_ptr() in scope.routed]
        if both:                       # a parameter with a routed sum AND an autograd gradient: add the latter
            torch._foreach_add_([v for v, _ in both], [g for _, g in both])
        for p, v in zip(self.trainable, self.flat.views):
            p.grad = v
        # loss and metric into the flat buffer's scalar slots and the report slot: ONE launch (three runtime copies of 4 bytes were three
        # launches on the back-end's chain)
        l0, a0 = loss.detach().reshape(1), acc.detach().reshape(1).to(torch.float32)
        _copy_many([self.flat.scalars[0:1], self.flat.scalars[1:2], self.loss.reshape(1)], [l0, a0, l0])

    def _update(self):
        if isinstance(self.opt, FlatAdamW):
            self.opt.step()            # clipping is fused into the flat update
            return
        if self.grad_clip is not None:
            torch.nn.utils.clip_grad_norm_(self.trainable, self.grad_clip, norm_type=2)
        self.opt.step()

    def _capture(self):
        keep = _TrainingState(self.model, self.opt, self.flat.flat)     # the warm-up below must leave no trace (see _TrainingState)
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):                 # warm-up on a side stream, as graph capture requires
            for _ in range(2):
                self._forward_backward()
                self._update()
        torch.cuda.current_stream(self.device).wait_stream(s)
        torch.cuda.synchronize(self.device)
        self._g_fb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g_fb):
            self._forward_backward()
        self._g_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g_opt):
            self._update()
        keep.restore()
        if isinstance(self.opt, FlatAdamW):
            self.opt.captured = True
        torch.cuda.synchronize(self.device)

    def step(self, pts=None, labels=None):
        if pts is not None:
            self.pts.copy_(pts)
        if labels is not None:
            self.labels.copy_(labels)
        if self.use_graph:
            if self._g_fb is None:
                self._capture()
            self._g_fb.replay()
            if self.distributed:
                self.flat.reduce()
            self._g_opt.replay()
        else:
            self._forward_backward()
            if self.distributed:
                self.flat.reduce()
            self._update()
        return self.loss

    def step_inputs(self, *tensors):
        """One step of a `loss_fn` recipe on new input tensors (copied into the static buffers)."""
        for dst, src in zip(self.inputs, tensors):
            dst.copy_(src)
        return self.step()


def _runs_beside(a, b, cycles=400000):
    """Do kernels on streams a and b run side by side?  torch hands out streams from a pool of 32 and the runtime maps streams onto a
    handful of hardware queues: two streams on ONE queue run their kernels one after the other, and a two-stream pipeline on them is the
    one-stream step (measured: 6.05 instead of 4.18 ms, depending only on how many streams the process had created before).  Two
    single-thread spin kernels (torch.cuda._sleep), timed alone and together."""
    def timed(streams):
        torch.cuda.synchronize(a.device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(a)
        for s_ in streams:
            if s_ is not a:
                s_.wait_event(e0)
            with torch.cuda.stream(s_):
                torch.cuda._sleep(cycles)
        for s_ in streams:
            if s_ is not a:
                a.wait_stream(s_)
        e1.record(a)
        torch.cuda.synchronize(a.device)
        return e0.elapsed_time(e1)
    timed([a, b])                    # (first use of a stream creates its queue)
    one = min(timed([a]), timed([a]))
    both = min(timed([a, b]), timed([a, b]))
    return both < 1.5 * one


def _concurrent_stream(device, tries=12):
    """A stream whose kernels run beside those of the CURRENT stream (the step driver replays its back-end graph there)."""
    cur = torch.cuda.current_stream(device)
    s = None
    for _ in range(tries):
        s = torch.cuda.Stream(device=device)
        if _runs_beside(cur, s):
            return s
    import warnings
    warnings.warn("upp_hip: no stream that runs beside the current one was found in %d tries; the pipelined step will run its halves one after the other" % tries)
    return s


_BACK_END_KEYS = ('downstream', 'bnorm', 'cls_')     # trainable parameters the prompting front-end never reads


class PipelinedTrainStep(TrainStep):
    """TrainStep with everything that reads no trainable parameter (rectify + completion prompters, grouping and patch
    embedding of the prompted cloud: `model.prompt_tokens`) software-pipelined against the trainable back-end
    (`model.forward_tokens`): while stream A runs forward + loss + backward of batch k, stream B already
    turns batch k+1 into prompted clouds.  Both halves are launch/latency-bound at B = 32, so two HIP graphs on two
    streams overlap almost completely (8.0 -> ~6 ms per step).  Results are those of the sequential order
    front(k+1), back(k), opt(k):

      * the front-end reads no trainable parameter (checked: every trainable name must match _BACK_END_KEYS), so the
        prompted clouds are what the sequential step would compute;
      * state both halves could write: the running statistics of the patch-embedding BatchNorms.  `prompt_tokens` covers
        every call of the patch embedding, so the back-end does not touch them; should a model's back-end do so, they are
        redirected to zero-initialised shadow buffers during the back-end (a momentum update from 0 leaves m * batch
        statistic there) and  real = (1 - m) * real + shadow  applies the update after the join, in that order, exactly
        (the warm-up detects whether this merge is needed).

    step(pts, labels) feeds batch k and returns the loss of batch k-1 (one step of latency; the first call only primes the
    pipeline; flush() finishes the last batch).  The front graph is captured on its own stream so that library GEMM
    workspaces are not shared between graphs that run concurrently."""

    def __init__(self, model, batch_shape, grad_clip=10.0, forward_kwargs=None, lr=5e-4, front_fn=None, back_fn=None, extras=None,
                 back_end_keys=_BACK_END_KEYS):
        """Other recipes than the classification one: front_fn(model, pts) -> tuple of tensors (the hand-over state; default
        model.prompt_tokens), back_fn(model, state, *extras) -> (loss, metric) (default: cross-entropy of
        model.forward_tokens(*state) against extras[0] = labels), extras = example tensors of the per-batch inputs the
        back-end needs besides the state (labels, targets, ...; one static copy per pipeline slot);
        back_end_keys = substrings every trainable parameter name must contain one of (the front-end must be frozen)."""
        super().__init__(model, batch_shape, grad_clip=grad_clip, use_graph=True, forward_kwargs=forward_kwargs, lr=lr)
        if self.device.type != 'cuda' or not (hasattr(model, 'prompt_tokens') or front_fn is not None):
            raise RuntimeError("PipelinedTrainStep needs a HIP device and a model with prompt_tokens() / forward_tokens()")
        names = {id(p): n for n, p in model.named_parameters()}
        bad = [names[id(p)] for p in self.trainable if not any(k in names[id(p)] for k in back_end_keys)]
        if bad:
            raise RuntimeError("front-end / back-end pipelining needs a frozen front-end; trainable: %s ..." % bad[:3])
        self.front_fn, self.back_fn = front_fn, back_fn
        if back_fn is not None and extras is None:
            raise RuntimeError("back_fn needs extras (example tensors of its per-batch inputs; may be an empty list)")
        if not (self.kw.get('completion_prompt') or self.kw.get('denoise')):
            raise RuntimeError("nothing to pipeline: both prompters are off")
        self.point_num = self.kw.get('point_num', 1024)
        self.kw_back = dict(self.kw, completion_prompt=False, denoise=False)
        B = batch_shape[0]
        with torch.no_grad():       # shapes / dtypes of the hand-over state: one run of the front-end on a random batch
            was = model.training
            x0 = torch.zeros(batch_shape, device=self.device).uniform_(-1, 1)
            if front_fn is not None:
                probe = front_fn(model.eval(), x0)
            else:
                probe = model.eval().prompt_tokens(x0, completion_prompt=False, denoise=False, point_num=self.point_num)
            model.train(was)
        self.state = [[torch.zeros_like(t) for t in probe] for _ in range(2)]
        self.labels2 = [torch.zeros(B, dtype=torch.long, device=self.device) for _ in range(2)]
        self.extras2 = [[t.detach().clone().to(self.device) for t in (extras or [])] for _ in range(2)]
        # (stream priorities do not help: round 2 5.89-5.92 ms for -1 / 0; round 5: front-end stream at high priority 4.58 against 4.54-4.55 ms,
        #  the whole driver on a high-priority stream of its own 4.83-4.84 against 4.63-4.64; the runtime offers no priority BELOW normal)
        self.s_front = _concurrent_stream(self.device)
        # UPP_PIPE_FPS_FORM (host-side A/B switch, tools/micro/fps_cpw_ab.sh): "0" = the spread form, "<clouds>,<exclusive>" = that packed form
        form = os.environ.get("UPP_PIPE_FPS_FORM", "2,1")
        self._fps_form = (1, False) if form == "0" else (int(form.split(",")[0]), form.split(",")[1:] == ["1"])
        from models import upp_layers as _L
        self._L = _L
        self._gen_front = torch.Generator(device=self.device)          # the front-end's own random stream (see upp_layers.use_rng)
        self._gen_front.manual_seed(torch.initial_seed() + 0x5EED)
        self._bank_front = _L.UniformBank(self._gen_front)
        self._bns = [m for m in model.encoder.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.track_running_stats]
        if any(m.momentum is None for m in self._bns):
            raise RuntimeError("cumulative-average BatchNorm statistics cannot be pipelined")
        self._real = [t for m in self._bns for t in (m.running_mean, m.running_var)]
        self._shadow = [torch.zeros_like(t) for t in self._real]
        self._decay = [1.0 - m.momentum for m in self._bns for _ in range(2)]
        self._counters = [m.num_batches_tracked for m in self._bns]
        self._shadow_counters = [torch.zeros_like(t) for t in self._counters]
        self._k = 0
        self._merge = True
        self._g_front = self._g_back = None
        self._ev_front = [torch.cuda.Event() for _ in range(2)]
        self._ev_back = [torch.cuda.Event() for _ in range(2)]
        self._ev_input = torch.cuda.Event()

    # -- the three parts of a step -----------------------------------------------------------
    def _front(self, p):
        # (the front-end runs beside the back-end: its FPS launches keep to CUs of their own -- ops.fps_form; UPP_PIPE_FPS_FORM=0: the spread form)
        with torch.no_grad(), self._L.use_rng(self._bank_front), HF.ops.fps_form(*self._fps_form):
            if self.front_fn is not None:
                state = self.front_fn(self.model, self.pts)
            else:
                state = self.model.prompt_tokens(self.pts, completion_prompt=bool(self.kw.get('completion_prompt')),
                                                 denoise=bool(self.kw.get('denoise')), point_num=self.point_num)
            state = list(state)
            if all(s_.is_cuda and s_.is_contiguous() and d.is_contiguous() and s_.dtype == d.dtype and s_.shape == d.shape
                   for d, s_ in zip(self.state[p], state)):
                HF.ops.copy_batched(self.state[p], state)         # hand-over buffers of this parity: one launch for all 16 tensors
            else:
                torch._foreach_copy_(self.state[p], state)

    class _Shadowed:
        def __init__(self, ts):
            self.ts = ts

        def __enter__(self):
            ts = self.ts
            it, ic = iter(ts._shadow), iter(ts._shadow_counters)
            for m in ts._bns:
                m._buffers['running_mean'], m._buffers['running_var'] = next(it), next(it)
                m._buffers['num_batches_tracked'] = next(ic)

        def __exit__(self, *exc):
            ts = self.ts
            it, ic = iter(ts._real), iter(ts._counters)
            for m in ts._bns:
                m._buffers['running_mean'], m._buffers['running_var'] = next(it), next(it)
                m._buffers['num_batches_tracked'] = next(ic)
            return False

    def _back(self, p):
        with PipelinedTrainStep._Shadowed(self):       # (the back-end no longer runs the patch embedding: kept as a guard)
            if self.back_fn is not None:
                self.loss_fn = lambda m: self.back_fn(m, tuple(self.state[p]), *self.extras2[p])
                self.inputs = []
                try:
                    self._forward_backward()
                finally:
                    self.loss_fn = None
            else:
                self._forward_backward(tuple(self.state[p]), self.labels2[p])

    def _tail(self):
        self._update()
        if not self._merge:
            return
        # fold the back-end's BatchNorm statistics (m * batch statistic, accumulated from zero) into the real buffers
        torch._foreach_mul_(self._real, self._decay)
        torch._foreach_add_(self._real, self._shadow)
        torch._foreach_zero_(self._shadow)
        torch._foreach_add_(self._counters, self._shadow_counters)
        torch._foreach_zero_(self._shadow_counters)

    def _capture(self):
        cur = torch.cuda.current_stream(self.device)
        keep = _TrainingState(self.model, self.opt, self.flat.flat)     # the warm-up must leave no trace (see _TrainingState)
        keep.tensors += self._shadow + self._shadow_counters
        keep.saved += [t.detach().clone() for t in self._shadow + self._shadow_counters]
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(cur)
        with torch.cuda.stream(s):                 # eager warm-up (also tunes GEMM shapes, sizes the uniform bank)
            for it in range(2):
                self._front(0)
                self._back(0)
                if it == 0:
                    # does the back-end touch the BatchNorms it shares with the front-end at all?  (With prompt_tokens
                    # covering the patch embedding it does not; the shadow merge then must not decay the real buffers.)
                    self._merge = any(int(c) != 0 for c in self._shadow_counters)
                self._tail()
        cur.wait_stream(s)
        torch.cuda.synchronize(self.device)
        self._g_front, self._g_back = [], []
        for p in range(2):
            gf = torch.cuda.CUDAGraph()
            gf.register_generator_state(self._gen_front)
            with torch.cuda.graph(gf, stream=self.s_front):
                self._front(p)
            gb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gb):
                self._back(p)
            self._g_front.append(gf)
            self._g_back.append(gb)
        self._g_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g_opt):
            self._tail()
        keep.restore()
        if isinstance(self.opt, FlatAdamW):
            self.opt.captured = True
        torch.cuda.synchronize(self.device)

    def _finish(self, p):
        if self.distributed:
            self.flat.reduce()
        self._g_opt.replay()

    def step(self, pts=None, labels=None, extras=None):
        """Feed batch k (pts, and labels or -- for a back_fn recipe -- its extras) and run the back-end of batch k-1."""
        if self._g_front is None:
            self._capture()
        p = self._k & 1
        cur = torch.cuda.current_stream(self.device)
        if self._merge:
            # the two halves share BatchNorm state that the tail folds: keep the strict fork / join per step
            if pts is not None:
                self.pts.copy_(pts)
            self._set_labels(p, labels, extras)
            self.s_front.wait_stream(cur)
            with torch.cuda.stream(self.s_front):
                self._g_front[p].replay()              # batch k: raw -> prompted[p]
            if self._k > 0:
                self._g_back[1 - p].replay()           # batch k-1: forward + loss + backward, concurrently
            cur.wait_stream(self.s_front)
            if self._k > 0:
                self._finish(1 - p)
            self._k += 1
            return self.loss
        # Loose coupling: the only edges are the data ones.  front(k) waits for its input copy and for back(k-2), the last
        # reader of the hand-over buffers of this parity; back(k-1) waits for front(k-1).  The front-end of the next batch
        # therefore also runs beside the gradient all-reduce and the optimizer of the previous one (nothing they touch is
        # read by the front-end: checked at construction and by the warm-up), not only beside its back-end.
        if self._k > 0:
            cur.wait_event(self._ev_front[1 - p])      # front(k-1) done: its state is complete, and self.pts is free again
        if pts is not None:
            self.pts.copy_(pts)
        self._set_labels(p, labels, extras)
        self._ev_input.record(cur)
        self.s_front.wait_event(self._ev_input)
        if self._k > 1:
            self.s_front.wait_event(self._ev_back[p])  # back(k-2) has read state[p]
        with torch.cuda.stream(self.s_front):
            self._g_front[p].replay()                  # batch k: raw -> prompted[p]
            self._ev_front[p].record(self.s_front)
        if os.environ.get("UPP_PIPE_SERIAL"):          # (diagnostic, tools/micro/pipe_race_probe.py: the two halves one after the other)
            cur.wait_stream(self.s_front)
        if self._k > 0:
            self._g_back[1 - p].replay()               # batch k-1: forward + loss + backward
            self._ev_back[1 - p].record(cur)
            self._finish(1 - p)
        self._k += 1
        return self.loss

    def _set_labels(self, p, labels, extras=None):
        if extras is not None:            # (without: the slot keeps what it holds -- the construction-time examples at first)
            for dst, src in zip(self.extras2[p], extras):
                dst.copy_(src)
        if labels is not None:
            self.labels2[p].copy_(labels)
            self.labels.copy_(labels)
        elif self._k < 2:
            self.labels2[p].copy_(self.labels)

    def flush(self):
        """Run the back-end of the batch whose front-end ran last (end of an epoch)."""
        if self._k > 0:
            p = (self._k - 1) & 1
            torch.cuda.current_stream(self.device).wait_stream(self.s_front)
            self._g_back[p].replay()
            self._finish(p)
            self.s_front.wait_stream(torch.cuda.current_stream(self.device))
        self._k = 0
        return self.loss
