"""HashNet (tests/refspec.py) evaluated with torch int64 ops on the device, batched.
Bit-identical to the python / C statements: every intermediate stays below 2^62."""
import torch

import refspec as R

P31 = R.P31


def _g(h, i):
    x = (h + i * 1000003) % P31
    x = (x * 48271 + 7) % P31
    x = x ^ (x >> 15)
    x = (x * 69621 + 3) % P31
    return x


class HashNetTorch:
    """duck-typed network: initial_inference(image) / recurrent_inference(hidden, action) on batches"""

    def __init__(self, mode="hash", device="cuda"):
        self.mode = mode
        self.device = torch.device(device)
        self.pow2 = torch.tensor([2.0 ** -e for e in range(41)], dtype=torch.float64, device=self.device)
        self.idx = torch.arange(65, dtype=torch.int64, device=self.device)
        self.w_img = ((torch.arange(576, dtype=torch.int64, device=self.device) + 1) * 2654435761) % P31
        self.w_act = torch.arange(128, dtype=torch.int64, device=self.device) + 1
        self.hid_i = torch.arange(512, dtype=torch.int64, device=self.device) + 1

    def outputs(self, h):
        n = h.shape[0]
        if self.mode == "uniform":
            return (torch.zeros(n, 1, device=self.device), torch.full((n, 65), 1.0 / 64.0, device=self.device))
        if self.mode == "const":
            return (torch.full((n, 1), 0.25, device=self.device), torch.ones(n, 65, device=self.device))
        v_int = _g(h, 0) % 8193 - 4096
        if self.mode == "sat":
            v = torch.where((v_int & 1) == 1, 1.0, -1.0).to(torch.float32)
        else:
            v = (v_int.double() / 4096.0).float()
        hh = h[:, None]
        m = _g(hh, self.idx[None, :] + 1) % 8388608
        e = _g(hh, self.idx[None, :] + 101) % 41
        pol = ((1.0 + m.double() / 8388608.0) * self.pow2[e]).float()
        return v.view(n, 1), pol

    def hidden(self, h):
        hid = (h[:, None] * self.hid_i[None, :]) % 65521
        hid[:, 0] = h % 65536
        hid[:, 1] = h // 65536
        return hid.float().view(-1, 8, 8, 8)

    def h_from_hidden(self, hidden):
        flat = hidden.reshape(hidden.shape[0], -1)
        return flat[:, 0].long() + 65536 * flat[:, 1].long()

    def check_hidden(self, hidden):
        """every one of the 512 entries must be the fill of the h it carries (catches partial copies)"""
        h = self.h_from_hidden(hidden)
        return bool((self.hidden(h) == hidden.reshape(-1, 8, 8, 8).float()).all())

    def initial_inference(self, image):
        h = (image.reshape(image.shape[0], -1).long() * self.w_img[None, :]).sum(1) % P31
        v, p = self.outputs(h)
        return v, p, self.hidden(h)

    def recurrent_inference(self, hidden, action):
        h = self.h_from_hidden(hidden)
        code = (action.reshape(action.shape[0], -1).long() * self.w_act[None, :]).sum(1)
        x = (h + code * 7919) % P31
        x = (x * 48271 + 11) % P31
        x = x ^ (x >> 13)
        h2 = (x * 69621 + 1) % P31
        v, p = self.outputs(h2)
        return v, p, self.hidden(h2)
