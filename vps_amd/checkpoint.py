"""`load_checkpoint` with the semantics tools/test_vpq.py:135-137 relies on (mmcv.runner.load_checkpoint of mmcv 0.2.x, a third-party
dependency that is not part of the reference tree: restated from its published behaviour).

    checkpoint = load_checkpoint(model, 'work_dirs/cityscapes_vps/fusetrack_vpct/latest.pth', map_location='cpu')
    model.CLASSES = checkpoint['meta']['CLASSES']          # test_vpq.py:140-143

* the file is an OrderedDict of tensors, or a dict with a 'state_dict' entry (+ 'meta', 'optimizer'); anything else raises;
* keys saved from a (Distributed)DataParallel wrapper start with 'module.': the prefix is stripped when the FIRST key has it;
* a model that is itself a wrapper (`model.module`) is unwrapped;
* matching tensors are copied IN PLACE into the model's own state; a shape mismatch raises RuntimeError naming the parameter;
* `strict=False` (the default, what test_vpq.py uses): unexpected and missing keys are reported (logger.warn / print), not fatal;
  `strict=True` raises with the same message;
* returns the loaded checkpoint object.
In addition the packed HIP weights of the model are invalidated (an in-place copy does not go through `load_state_dict`, whose
post-hook does that for the torch API), and the report is returned as `checkpoint['_vps_load_report']` for tools/run_config3.py.
"""
from collections import OrderedDict

import torch


def load_state_dict(module, state_dict, strict=False, logger=None):
    unexpected_keys = []
    own_state = module.state_dict()
    for name, param in state_dict.items():
        if name not in own_state:
            unexpected_keys.append(name)
            continue
        if isinstance(param, torch.nn.Parameter):
            param = param.data
        try:
            own_state[name].copy_(param)
        except Exception:
            raise RuntimeError('While copying the parameter named {}, whose dimensions in the model are {} and whose dimensions in the '
                               'checkpoint are {}.'.format(name, own_state[name].size(), param.size()))
    missing_keys = sorted(set(own_state.keys()) - set(state_dict.keys()))
    err_msg = []
    if unexpected_keys:
        err_msg.append('unexpected key in source state_dict: {}\n'.format(', '.join(unexpected_keys)))
    if missing_keys:
        err_msg.append('missing keys in source state_dict: {}\n'.format(', '.join(missing_keys)))
    err_msg = '\n'.join(err_msg)
    if err_msg:
        if strict:
            raise RuntimeError(err_msg)
        elif logger is not None:
            logger.warn(err_msg)
        else:
            print(err_msg)
    if hasattr(module, 'invalidate'):
        module.invalidate()
    return dict(unexpected=unexpected_keys, missing=missing_keys, loaded=len(state_dict) - len(unexpected_keys))


def load_checkpoint(model, filename, map_location=None, strict=False, logger=None):
    checkpoint = torch.load(filename, map_location=map_location)
    if isinstance(checkpoint, OrderedDict):
        state_dict = checkpoint
    elif isinstance(checkpoint, dict) and 'state_dict' in checkpoint:
        state_dict = checkpoint['state_dict']
    else:
        raise RuntimeError('No state_dict found in checkpoint file {}'.format(filename))
    if len(state_dict) and list(state_dict.keys())[0].startswith('module.'):
        state_dict = OrderedDict((k[7:], v) for k, v in state_dict.items())
    target = model.module if hasattr(model, 'module') else model
    report = load_state_dict(target, state_dict, strict, logger)
    if isinstance(checkpoint, dict):
        checkpoint['_vps_load_report'] = report
    return checkpoint
