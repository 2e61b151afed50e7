"""Base class of the HIP-backed components.

Each component is a torch.nn.Module whose sub-modules (nn.Conv2d, nn.BatchNorm2d, nn.Linear, ...) are used purely
as PARAMETER CONTAINERS with the reference's attribute names, so the reference's checkpoints (`state_dict` keys
listed in SURVEY §8b) load unchanged. Their torch forward() is never executed: `prepare()` packs the weights for
libvpship (K-major, BN folded, transposed-conv parity classes) and `run()` launches the HIP kernels on NHWC maps.
"""
import torch
import torch.nn as nn


class HipModule(nn.Module):
    def __init__(self):
        super().__init__()
        self._packed_device = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate())

    def invalidate(self):
        """weights changed (load_state_dict / .to): repack lazily at the next run."""
        self._packed_device = None
        for m in self.children():
            if isinstance(m, HipModule):
                m.invalidate()

    def ensure_packed(self, device):
        device = torch.device(device)
        if self._packed_device != device:
            self.pack(device)
            self._packed_device = device

    def pack(self, device):  # pragma: no cover - overridden
        raise NotImplementedError

    def init_weights(self, pretrained=None):
        """kept for API compatibility with the reference builders; weights come from checkpoints or synth.py"""
        return None
