"""Generate tests/golden/*.npz by running the REFERENCE's own model code (oracle/ref_shim.py)
in the build container.  Committed together with its outputs; never runs on the GPU box.

    python oracle/gen_golden.py

Fixtures (all eval mode, key-seeded weights from tests/_seeded.py):
  upp_model.npz    end-to-end logits of Point_MAE_unify: clean (B,1024,3) and noisy-train
                   (B,1096,3, denoise + completion prompters), CE loss and gradients of a few
                   PEFT-trainable tensors (+ L2 norm of every trainable gradient)
  upp_seg.npz      Point_MAE_unify_seg (unify_shapenetpart_seg.yaml): log-probabilities of a noisy (2,1624,3) cloud
                   at 2048 label points (first 256 points + per-point sums / argmax of all) and of a clean run, NLL loss
  point_mae.npz    Point_MAE (pretrain.yaml): Chamfer-L2 loss and gradient norms for a fixed random mask
  pretask.npz      Point_MAE_pretask_dev (pretask.yaml) in TRAIN mode with every dropout / drop-path probability set to 0:
                   predicted centres, rebuilt points, noise loss, recall, the three Chamfer-L1 terms of the pre-task recipe
                   (tools/runner_pretask.py:220-225) and the gradient norm of every parameter
  upp_stage2.npz   the stage-2 ("joint optimisation", tools/runner_module.py:230-244) step: logits, loss, gradient norms of every
                   stage-2 trainable tensor and 14 full gradient arrays (prompter heads, mask token, rectify prompter)
  upp_seg_train.npz  one TRAIN-mode step of the part-segmentation recipe (batch-statistics BatchNorm, dropout 0): log-probabilities, loss,
                   gradient norms of all trainable tensors, 13 full gradient arrays, sampled rows / columns of the large head weights
  upp_modules.npz  per-module input/output pairs: Encoder, Attention, Block (downstream path
                   with prompt propagation incl. the index-stride behaviour), TransformerDecoder,
                   RectifyPrompter, propagate, Group index outputs
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

import _seeded  # noqa: E402
import ref_shim  # noqa: E402

PEFT_KEYS = ['downstream_adapter', 'downstream_adapter1', 'downstream_prompts', 'bnorm', 'cls_pos', 'cls_token',
             'cls_head_finetune']  # reference tools/runner_module.py:62-66


def deterministic_train_mode(model):
    """train() with all stochastic layers neutralised (shared by the fixture generator and the tests)."""
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, 'drop_prob'):
            m.drop_prob = 0.0
    return model


def pretask_inputs():
    gt = _seeded.unit_ball_clouds(2, 1280, seed=31)
    partial, cropping = gt[:, :1024].contiguous(), gt[:, 1024:].contiguous()
    noise = _seeded.noisy_clouds(2, 1024, seed=32)[:, 1024:1076].contiguous()      # 52 outlier points
    return gt, partial, cropping, torch.cat([partial, noise], dim=1).contiguous()


def gen_pretask(R, out_dir):
    model = R.MODELS.build(ref_shim.model_cfg('pretask'))
    deterministic_train_mode(_seeded.fill(model))
    gt, partial, cropping, points = pretask_inputs()
    from extensions.chamfer_dist import ChamferDistanceL1
    cd = ChamferDistanceL1()
    center, rebuild, noise_loss, recall = model(points, point_num=1024, train_with_gaussian=True, predict_center_num=16)
    coarse, crop_dense, dense = cd(center, cropping), cd(rebuild, cropping), cd(torch.cat([partial, rebuild], dim=1), gt)
    loss = coarse + crop_dense + dense + noise_loss
    loss.backward()
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    names = sorted(grads)
    model.eval()
    with torch.no_grad():
        center_eval, rebuild_eval = model(partial, point_num=1024, train_with_gaussian=False)
    np.savez_compressed(os.path.join(out_dir, "pretask.npz"), center=center.detach().numpy(), rebuild=rebuild.detach().numpy(),
                        noise_loss=noise_loss.detach().numpy(), recall=recall.numpy(), coarse=coarse.detach().numpy(),
                        crop_dense=crop_dense.detach().numpy(), dense=dense.detach().numpy(), loss=loss.detach().numpy(),
                        grad_names=np.array(names), grad_norms=np.array([grads[n].norm().item() for n in names]),
                        center_eval=center_eval.numpy(), rebuild_eval_head=rebuild_eval[:, :128].numpy(),
                        n_params=np.array(sum(p.numel() for p in model.parameters())), n_keys=np.array(len(model.state_dict())))
    print("pretask loss", loss.item(), "noise", noise_loss.item(), "recall", recall.item(), "params", sum(p.numel() for p in model.parameters()),
          "keys", len(model.state_dict()), "grads", len(names))


STAGE2_KEYS = ['downstream_adapter', 'downstream_adapter1', 'downstream_prompts', 'dense_pred', 'mask_token', 'rectify_prompter',
               'shape_pred', 'coarse_pred', 'predict_token_generator', 'mask_prompter', 'mask_token_generator']
# reference tools/runner_module.py:232-238 ("joint optimisation" from epoch args.joint_optimization on: the prompter heads train,
# the classification head / bnorm / cls tokens of stage 1 are frozen again)
STAGE2_KEEP = ['mask_token', 'dense_pred.0.weight', 'dense_pred.0.bias', 'coarse_pred.2.bias', 'coarse_pred.0.weight', 'shape_pred.0.weight',
               'shape_pred.2.bias', 'predict_token_generator.2.weight', 'rectify_prompter.score_head.3.weight',
               'rectify_prompter.score_head.0.bias', 'rectify_prompter.propagation1.mlp_convs.0.weight',
               'rectify_prompter.abstraction.mlp_convs.2.weight', 'blocks.blocks.3.downstream_adapter.ln1.weight',
               'blocks.blocks.0.downstream_prompts']


def gen_stage2(R, out_dir):
    """upp_stage2.npz: the noisy-train step of the SECOND stage of the recipe -- the gradient reaches the prompter heads
    through the grouping / patch embedding of the prompted cloud, the FPS gather, rebuild_points, the frozen decoder and
    backbone paths and the rectification offsets.  Eval-mode layers (as upp_model.npz), reference classes, key-seeded weights."""
    model = R.MODELS.build(ref_shim.model_cfg())
    _seeded.fill(model).eval()
    for name, p in model.named_parameters():
        p.requires_grad_(any(k in name for k in STAGE2_KEYS))
    noisy = _seeded.noisy_clouds(2, 1024, seed=0)
    labels = torch.tensor([3, 17])
    logits = model(noisy, completion_prompt=True, denoise=True, point_num=1024)
    loss, _ = model.get_loss_acc(logits, labels)
    loss.backward()
    grads = {n: p.grad for n, p in model.named_parameters() if p.requires_grad and p.grad is not None}
    names = sorted(grads)
    np.savez_compressed(os.path.join(out_dir, "upp_stage2.npz"), labels=labels.numpy(), logits=logits.detach().numpy(),
                        loss=loss.detach().numpy(), grad_names=np.array(names),
                        grad_norms=np.array([grads[n].norm().item() for n in names], dtype=np.float64),
                        **{"grad::" + k: grads[k].numpy() for k in STAGE2_KEEP})
    print("stage2 loss", loss.item(), "trainable tensors with a gradient", len(names), "of",
          sum(1 for _, p in model.named_parameters() if p.requires_grad),
          "norms", {k: float(grads[k].norm()) for k in STAGE2_KEEP[:6]})


SEG_PEFT = ['downstream_adapter', 'downstream_prompts', 'label_conv', 'propagation_0', 'seg_head', 'propagation_1']   # reference tools/runner_unify_seg.py:143-146
SEG_KEEP = ['seg_head.7.weight', 'seg_head.7.bias', 'seg_head.5.weight', 'seg_head.1.bias', 'propagation_0.mlp_bns.0.weight', 'propagation_0.mlp_bns.1.bias',
            'propagation_0.mlp_convs.1.bias', 'label_conv.0.weight', 'label_conv.3.weight', 'blocks.blocks.0.downstream_prompts',
            'blocks.blocks.5.downstream_prompts', 'blocks.blocks.0.downstream_adapter.ln1.weight', 'blocks.blocks.11.downstream_adapter.ln2.weight']
SEG_SAMPLED = ['seg_head.0.weight', 'seg_head.4.weight', 'propagation_0.mlp_convs.0.weight', 'propagation_0.mlp_convs.1.weight']   # every 16th row / column


def seg_train_inputs():
    spts = _seeded.noisy_clouds(2, 1552, seed=11)            # (2,1624,3): 1552 + 72 noise points
    lpts = _seeded.unit_ball_clouds(2, 2048, seed=12)
    onehot = torch.zeros(2, 16); onehot[0, 3] = 1; onehot[1, 11] = 1
    tgt = torch.randint(0, 50, (2, 2048), generator=torch.Generator().manual_seed(13))
    return spts, lpts, onehot, tgt


def gen_seg_train(R, out_dir):
    """upp_seg_train.npz: ONE training step of the part-segmentation recipe (reference tools/runner_unify_seg.py:143-152 parameter list,
    :190-261 step; models/Point_MAE_unify_segment.py:475-625) on the reference's class in TRAIN mode -- BatchNorm layers on batch
    statistics, every dropout / drop-path probability 0 -- : log-probabilities, NLL loss, the gradient norm of every trainable tensor,
    13 full gradient arrays and every 16th row / column of the four large head weights."""
    seg = R.MODELS.build(ref_shim.model_cfg('unify_shapenetpart_seg'))
    deterministic_train_mode(_seeded.fill(seg))
    for name, p in seg.named_parameters():
        p.requires_grad_(any(k in name for k in SEG_PEFT))
    spts, lpts, onehot, tgt = seg_train_inputs()
    logp = seg(spts, onehot, label_points=lpts, completion_prompt=True, denoise=True, point_num=1536)
    loss = seg.get_loss(logp.reshape(-1, 50), tgt.reshape(-1))
    loss.backward()
    grads = {n: p.grad for n, p in seg.named_parameters() if p.requires_grad and p.grad is not None}
    names = sorted(grads)
    keep = {"grad::" + k: grads[k].numpy() for k in SEG_KEEP}
    keep.update({"sampled::" + k: grads[k].squeeze(-1)[::16, ::16].contiguous().numpy() for k in SEG_SAMPLED})
    np.savez_compressed(os.path.join(out_dir, "upp_seg_train.npz"), logp_head=logp[:, :256].detach().numpy(),
                        logp_sum=logp.detach().double().sum(-1).numpy(), loss=loss.detach().numpy(), grad_names=np.array(names),
                        grad_norms=np.array([grads[n].norm().item() for n in names], dtype=np.float64), **keep)
    print("seg train loss", loss.item(), "trainable tensors with a gradient", len(names), "of",
          sum(1 for _, p in seg.named_parameters() if p.requires_grad), os.path.getsize(os.path.join(out_dir, "upp_seg_train.npz")) // 1024, "KiB")


class _keep_f64:
    """The reference calls `.float()` on coordinates in a few places (models/Point_MAE_pretask_dev.py:411); inside this scope a float64
    tensor stays float64 there, so that `.double()` models really evaluate in f64.  Generator-side only; the reference is not edited."""

    def __enter__(self):
        self.orig = torch.Tensor.float
        orig = self.orig
        torch.Tensor.float = lambda t, *a, **k: t if t.dtype == torch.float64 else orig(t, *a, **k)

    def __exit__(self, *exc):
        torch.Tensor.float = self.orig
        return False


def gen_seg_train_f64(R, out_dir):
    """upp_seg_train_f64.npz: the SAME step as gen_seg_train with the reference's class converted to float64 (`.double()`; the shimmed
    FPS / kNN take the f32 image of the coordinates, so the index sets are those of the f32 run) -- the arbiter for tests that compare
    two f32 evaluations of an ill-conditioned step (train-mode BatchNorm over 2 x 2048 rows, ReLU gates): loss, all gradient norms, the
    kept / sampled gradient arrays, as float64."""
    seg = R.MODELS.build(ref_shim.model_cfg('unify_shapenetpart_seg'))
    deterministic_train_mode(_seeded.fill(seg))
    for name, p in seg.named_parameters():
        p.requires_grad_(any(k in name for k in SEG_PEFT))
    seg = seg.double()
    spts, lpts, onehot, tgt = seg_train_inputs()
    with _keep_f64():
        logp = seg(spts.double(), onehot.double(), label_points=lpts.double(), completion_prompt=True, denoise=True, point_num=1536)
        loss = seg.get_loss(logp.reshape(-1, 50), tgt.reshape(-1))
        loss.backward()
    grads = {n: p.grad for n, p in seg.named_parameters() if p.requires_grad and p.grad is not None}
    assert all(g.dtype == torch.float64 for g in grads.values()) and logp.dtype == torch.float64
    names = sorted(grads)
    keep = {"grad::" + k: grads[k].numpy() for k in SEG_KEEP}
    keep.update({"sampled::" + k: grads[k].squeeze(-1)[::16, ::16].contiguous().numpy() for k in SEG_SAMPLED})
    path = os.path.join(out_dir, "upp_seg_train_f64.npz")
    np.savez_compressed(path, logp_head=logp[:, :256].detach().numpy(), loss=loss.detach().numpy(), grad_names=np.array(names),
                        grad_norms=np.array([grads[n].norm().item() for n in names], dtype=np.float64), **keep)
    print("seg train (f64) loss", loss.item(), "tensors", len(names), os.path.getsize(path) // 1024, "KiB")


def gen_stage2_f64(R, out_dir):
    """upp_stage2_f64.npz: gen_stage2 with the reference's class in float64 (same index sets): loss, logits, gradient norms and the kept
    arrays as float64 -- the arbiter for the geometry-path gradients, where two f32 evaluations differ through max-pool arg-max flips."""
    model = R.MODELS.build(ref_shim.model_cfg())
    _seeded.fill(model).eval()
    for name, p in model.named_parameters():
        p.requires_grad_(any(k in name for k in STAGE2_KEYS))
    model = model.double()
    with _keep_f64():
        logits = model(_seeded.noisy_clouds(2, 1024, seed=0).double(), completion_prompt=True, denoise=True, point_num=1024)
        loss, _ = model.get_loss_acc(logits, torch.tensor([3, 17]))
        loss.backward()
    grads = {n: p.grad for n, p in model.named_parameters() if p.requires_grad and p.grad is not None}
    assert all(g.dtype == torch.float64 for g in grads.values())
    names = sorted(grads)
    path = os.path.join(out_dir, "upp_stage2_f64.npz")
    keep = {}
    for k in STAGE2_KEEP:                                  # arrays above 64 KiB: every 4th row / column (the norms cover the rest)
        a = grads[k].numpy()
        if a.nbytes > 65536:
            keep["sampled4::" + k] = np.ascontiguousarray(a[::4, ::4])
        else:
            keep["grad::" + k] = a
    np.savez_compressed(path, logits=logits.detach().numpy(), loss=loss.detach().numpy(), grad_names=np.array(names),
                        grad_norms=np.array([grads[n].norm().item() for n in names], dtype=np.float64), **keep)
    print("stage2 (f64) loss", loss.item(), "tensors", len(names), os.path.getsize(path) // 1024, "KiB")


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    R = ref_shim.load()
    if sys.argv[1:] == ['pretask']:
        gen_pretask(R, os.path.join(ROOT, "tests", "golden"))
        return
    if sys.argv[1:] == ['stage2']:
        gen_stage2(R, os.path.join(ROOT, "tests", "golden"))
        return
    if sys.argv[1:] == ['seg_train']:
        gen_seg_train(R, os.path.join(ROOT, "tests", "golden"))
        return
    if sys.argv[1:] == ['seg_train_f64']:
        gen_seg_train_f64(R, os.path.join(ROOT, "tests", "golden"))
        return
    if sys.argv[1:] == ['stage2_f64']:
        gen_stage2_f64(R, os.path.join(ROOT, "tests", "golden"))
        return
    cfg = ref_shim.model_cfg()
    model = R.MODELS.build(cfg)
    _seeded.fill(model).eval()
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)

    # ---------------- end to end
    B = 2
    clean = _seeded.unit_ball_clouds(B, 1024, seed=0)
    noisy = _seeded.noisy_clouds(B, 1024, seed=0)
    labels = torch.tensor([3, 17])
    for name, p in model.named_parameters():
        p.requires_grad_(any(k in name for k in PEFT_KEYS))
    logits_clean = model(clean, completion_prompt=False, denoise=False, point_num=1024)
    logits_noisy = model(noisy, completion_prompt=True, denoise=True, point_num=1024)
    loss, acc = model.get_loss_acc(logits_noisy, labels)
    loss.backward()
    grads = {n: p.grad for n, p in model.named_parameters() if p.requires_grad and p.grad is not None}
    keep = ['cls_token', 'cls_pos', 'blocks.blocks.0.downstream_prompts', 'blocks.blocks.5.downstream_adapter.ln1.weight',
            'blocks.blocks.11.downstream_adapter.ln2.bias', 'blocks.blocks.2.bnorm.weight', 'cls_head_finetune.8.bias',
            'cls_head_finetune.0.weight']
    names = sorted(grads)
    np.savez_compressed(
        os.path.join(out_dir, "upp_model.npz"),
        labels=labels.numpy(), logits_clean=logits_clean.detach().numpy(), logits_noisy=logits_noisy.detach().numpy(),
        loss=loss.detach().numpy(), grad_names=np.array(names),
        grad_norms=np.array([grads[n].norm().item() for n in names], dtype=np.float64),
        **{"grad::" + k: grads[k].numpy() for k in keep})
    print("logits_clean", logits_clean[0, :5].tolist(), "loss", loss.item(), "n_trainable", sum(g.numel() for g in grads.values()))

    # ---------------- per module
    mods = {}
    with torch.no_grad():
        g = torch.Generator().manual_seed(7)
        pts = _seeded.unit_ball_clouds(2, 1024, seed=3)
        grp = model.group_divider
        nb, center, idx, cidx = grp(pts, require_index=True, gather_idx=False)
        mods.update(group_pts=pts, group_neighborhood=nb, group_center=center, group_idx=idx, group_center_idx=cidx)
        tokens = model.encoder(nb)
        mods.update(encoder_out=tokens)
        x = torch.randn(2, 75, 384, generator=g)
        mods.update(attn_in=x, attn_out=model.blocks.blocks[0].attn(x))
        mods.update(mlp_out=model.blocks.blocks[0].mlp(x), adapter_out=model.blocks.blocks[0].downstream_adapter(x))
        # Block, downstream path with propagation (gather_idx False -> flat, batch-offset indices)
        lvl2 = R.uni.Group(num_group=32, group_size=8)
        _, c2, c1_idx, c2_idx = lvl2(center, require_index=True, gather_idx=False)
        xb = torch.randn(2, 65, 384, generator=g)
        kw = dict(path='downstream', downstream_adapter=True, downstream_prompts=True, classification=True,
                  center1=center, center1_idx=c1_idx, center2=c2, center2_idx=c2_idx, gather_idx=False,
                  prompt_propagation_after=True)
        mods.update(block_in=xb, block_center2=c2, block_c1_idx=c1_idx, block_c2_idx=c2_idx,
                    block0_out=model.blocks.blocks[0](xb, **kw), block7_out=model.blocks.blocks[7](xb, **kw))
        # same block with gather_idx=True indices (seg / pretask configs)
        _, c2g, c1g, c2ig = lvl2(center, require_index=True, gather_idx=True)
        kwg = dict(kw, center1_idx=c1g, center2_idx=c2ig, gather_idx=True)
        mods.update(block0_gather_out=model.blocks.blocks[0](xb, **kwg))
        xr = torch.randn(2, 32, 384, generator=g)
        mods.update(rect_in=xr, block1_rectify_out=model.blocks.blocks[1](xr, path='rectify', rectify_adapter=True,
                                                                          rectify_prompts=True, rectify_depth=3),
                    block4_pretask_out=model.blocks.blocks[4](xr, path='pretask', pretask_adapter=True,
                                                              pretask_prompts=True, pretask_depth=6))
        xd = torch.randn(2, 64, 384, generator=g)
        pd = torch.randn(2, 64, 384, generator=g)
        mods.update(dec_in=xd, dec_pos=pd, dec_out=model.MAE_decoder(xd, pd, 32, pretask_adapter=True, path='pretask'))
        npts = _seeded.noisy_clouds(2, 1024, seed=5)
        c32 = center[:, :32].contiguous()
        mods.update(rp_pts=npts, rp_center=c32, rp_tokens=xr, rp_out=model.rectify_prompter(npts, c32, xr))
        p2 = torch.randn(2, 32, 384, generator=g)
        p1 = torch.randn(2, 64, 384, generator=g)
        mods.update(prop_p1=p1, prop_p2=p2, prop_out=R.uni.propagate(center, c2, p1, p2, de_neighbors=8, dist_e=1e-3))
    np.savez_compressed(os.path.join(out_dir, "upp_modules.npz"),
                        **{k: (v.numpy() if torch.is_tensor(v) else v) for k, v in mods.items()})
    # ---------------- segmentation model (BASELINE config 5), eval mode
    seg = R.MODELS.build(ref_shim.model_cfg('unify_shapenetpart_seg'))
    _seeded.fill(seg).eval()
    spts = _seeded.noisy_clouds(2, 1552, seed=11)            # (2,1624,3): 1552 + 72 noise points
    lpts = _seeded.unit_ball_clouds(2, 2048, seed=12)
    onehot = torch.zeros(2, 16); onehot[0, 3] = 1; onehot[1, 11] = 1
    tgt = torch.randint(0, 50, (2, 2048), generator=torch.Generator().manual_seed(13))
    with torch.no_grad():
        logp = seg(spts, onehot, label_points=lpts, completion_prompt=True, denoise=True, point_num=1536)
        logp_clean = seg(lpts, onehot, label_points=None, completion_prompt=False, denoise=False, point_num=2048)
    loss = seg.get_loss(logp.reshape(-1, 50), tgt.reshape(-1))
    np.savez_compressed(os.path.join(out_dir, "upp_seg.npz"), onehot=onehot.numpy(), target=tgt.numpy(),
                        logp_head=logp[:, :256].numpy(), logp_sum=logp.double().sum(-1).numpy(), logp_argmax=logp.argmax(-1).numpy(),
                        logp_clean_head=logp_clean[:, :256].numpy(), loss=loss.numpy(),
                        n_params=np.array(sum(p.numel() for p in seg.parameters())), n_keys=np.array(len(seg.state_dict())))
    print("seg logp", tuple(logp.shape), "loss", loss.item(), "params", sum(p.numel() for p in seg.parameters()))
    # ---------------- Point-MAE pre-training model (Chamfer-L2 loss), eval-mode layers, fixed numpy mask
    mae = R.MODELS.build(ref_shim.model_cfg('pretrain'))
    _seeded.fill(mae).eval()
    mpts = _seeded.unit_ball_clouds(2, 1024, seed=21)
    np.random.seed(5)
    state = np.random.get_state()
    with torch.no_grad():
        nb_, cen_ = mae.group_divider(mpts)
        used_mask = mae.MAE_encoder._mask_center_rand(cen_)
    np.random.set_state(state)                    # the forward below redraws the same mask
    for p_ in mae.parameters():
        p_.requires_grad_(True)
    mloss = mae(mpts)
    mloss.backward()
    mg = {n: p_.grad for n, p_ in mae.named_parameters() if p_.grad is not None}
    feat = mae(mpts, eval=True) if False else None   # eval path needs .cuda() in the reference; not exercised
    mnames = sorted(mg)
    np.savez_compressed(os.path.join(out_dir, "point_mae.npz"), mask=used_mask.numpy(), loss=mloss.detach().numpy(),
                        grad_names=np.array(mnames), grad_norms=np.array([mg[n].norm().item() for n in mnames]),
                        g_mask_token=mg['mask_token'].numpy(), g_increase_bias=mg['increase_dim.0.bias'].numpy(),
                        n_params=np.array(sum(p_.numel() for p_ in mae.parameters())), n_keys=np.array(len(mae.state_dict())))
    print("point_mae loss", mloss.item(), "params", sum(p_.numel() for p_ in mae.parameters()), "masked", int(used_mask[0].sum()))
    gen_pretask(R, out_dir)
    gen_stage2(R, out_dir)
    gen_seg_train(R, out_dir)
    for f in ("upp_stage2.npz", "upp_model.npz", "upp_modules.npz", "upp_seg.npz", "point_mae.npz", "pretask.npz"):
        print(f, os.path.getsize(os.path.join(out_dir, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
