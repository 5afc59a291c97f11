"""HashNet behind the reference's network API (initial_inference / recurrent_inference on single
CPU tensors), evaluated with python ints / numpy.  Used by the golden generator to drive the
reference and by the compat tests to drive the shims -- the same object on both sides."""
import torch

import refspec as R


class HashNetwork:
    """duck-typed stand-in for network.Network: value (1,1), policy (1,65), hidden (1,8,8,8)"""

    def __init__(self, mode="hash"):
        self.mode = mode
        self.calls = []  # (parent_h, action_code)

    def _out(self, h):
        v, p = R.hashnet_outputs(h, self.mode)
        return (torch.tensor([[v]], dtype=torch.float32), torch.from_numpy(p).unsqueeze(0),
                torch.from_numpy(R.hashnet_hidden(h)).reshape(1, 8, 8, 8))

    def initial_inference(self, image):
        return self._out(R.hashnet_image_hash(image.cpu().numpy()))

    def recurrent_inference(self, hidden, action):
        h = R.hashnet_h_from_hidden(hidden.cpu().numpy())
        code = R.hashnet_action_code(action.cpu().numpy())
        self.calls.append((h, code))
        return self._out(R.hashnet_step(h, code))
