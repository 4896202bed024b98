"""Drop-in ``EfficientVRNet`` (reference: nets/efficient_vrnet.py:13-27).

Same constructor, ``forward(x, x_radar) -> (list[3] det maps, seg logits)`` and 887-key
``state_dict`` as the reference; the forward and backward passes run on the hand-written
gfx950 kernels of ``libvrnet_hip.so`` (see program.py).  There is no CPU / eager fallback:
without the library, or on a non-HIP tensor, ``forward`` raises.
"""
import torch
import torch.nn as nn

from .modules import WIDTH, DEPTH, CoCFpnDual, DecoupleHead


class EfficientVRNet(nn.Module):
    def __init__(self, num_classes, num_seg_classes, phi, *, img_size=(512, 512)):
        super().__init__()
        depth, width = DEPTH[phi], WIDTH[phi]          # `depth` is unused, as in the reference (:16-18)
        if isinstance(img_size, int):
            img_size = (img_size, img_size)
        self.phi, self.width, self.img_size = phi, width, tuple(img_size)
        self.num_classes, self.num_seg_classes = num_classes, num_seg_classes
        # "f32": every dense conv on the fp32 MFMA (the parity path, 1e-3 against the reference).  "bf16": operands of
        # the dense convs (activations, weights, output gradients) are rounded to bf16 when staged and multiplied on
        # the bf16 MFMA with fp32 accumulation -- BASELINE configs "bf16 with MFMA conv path"; tensors in HBM, norms,
        # clustering and every reduction stay fp32.  The reference's counterpart is torch.cuda.amp.autocast.
        self.compute_dtype = "f32"
        self.backbone = CoCFpnDual(width=width, num_seg_class=num_seg_classes, img_size=self.img_size)
        self.head = DecoupleHead(num_classes, width, depthwise=True)

    def forward(self, x, x_radar):
        from .program import run_forward        # imports the HIP library; raises if it is missing
        return run_forward(self, x, x_radar)
