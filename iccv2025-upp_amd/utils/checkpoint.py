"""Checkpoint wire format of the reference training tools (tools/builder.py:91-166) -- same file layout, so that
checkpoints written by either side load in the other:

    {'base_model': state_dict (keys optionally prefixed 'module.' by DistributedDataParallel),
     'optimizer' : optimizer.state_dict(),
     'epoch'     : int [, 'metrics': ..., 'best_metrics': ...]}

`load_model` also accepts the {'model': ...} layout of third-party backbones (builder.py:148-152).  Backbone checkpoints
(Point-MAE / ReCon / ... '.pth' files and UPP `prompter_bases/*.pth`) go through the models' own `load_model_from_ckpt`
(key rewrites of models/Point_MAE_unify.py:505-536)."""
import os

import torch


def _strip(sd):
    return {k.replace("module.", ""): v for k, v in sd.items()}


def save_checkpoint(base_model, optimizer, epoch, path, metrics=None, best_metrics=None, is_main=True):
    """builder.py:131-140.  Only the main rank writes.  FlatAdamW (upp_hip.train) exposes the same state_dict contract."""
    if not is_main:
        return None
    module = getattr(base_model, 'module', base_model)
    blob = {'base_model': module.state_dict(), 'optimizer': optimizer.state_dict() if optimizer is not None else {}, 'epoch': int(epoch)}
    if metrics is not None:
        blob['metrics'] = metrics
    if best_metrics is not None:
        blob['best_metrics'] = best_metrics
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(blob, path)
    return path


def load_model(base_model, ckpt_path, strict=True):
    """builder.py:142-166 -> (epoch, metrics).  Raises like the reference on a missing file / unknown layout."""
    if not os.path.exists(ckpt_path):
        raise NotImplementedError('no checkpoint file from path %s...' % ckpt_path)
    blob = torch.load(ckpt_path, map_location='cpu')
    if blob.get('model') is not None:
        sd = _strip(blob['model'])
    elif blob.get('base_model') is not None:
        sd = _strip(blob['base_model'])
    else:
        raise RuntimeError('mismatch of ckpt weight')
    getattr(base_model, 'module', base_model).load_state_dict(sd, strict=strict)
    metrics = blob.get('metrics', 'No Metrics')
    if not isinstance(metrics, (dict, str)):
        metrics = metrics.state_dict()
    return blob.get('epoch', -1), metrics


def resume_model(base_model, experiment_path, map_location='cpu'):
    """builder.py:91-114 -> (start_epoch, best_metrics); (0, 0) when there is no ckpt-last.pth."""
    path = os.path.join(experiment_path, 'ckpt-last.pth')
    if not os.path.exists(path):
        return 0, 0
    blob = torch.load(path, map_location=map_location)
    getattr(base_model, 'module', base_model).load_state_dict(_strip(blob['base_model']), strict=True)
    best = blob.get('best_metrics', {})
    if not isinstance(best, dict):
        best = best.state_dict()
    return blob['epoch'] + 1, best


def resume_optimizer(optimizer, experiment_path):
    """builder.py:116-129."""
    path = os.path.join(experiment_path, 'ckpt-last.pth')
    if not os.path.exists(path):
        return False
    optimizer.load_state_dict(torch.load(path, map_location='cpu')['optimizer'])
    return True
