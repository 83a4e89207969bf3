"""Anchor generators of the rotated heads (core/anchor/ranchor_generator.py:7-62).

The reference derives both from mmdet 2.19's ``AnchorGenerator`` (third-party, not under the
reference tree); the parts of it the rotated heads use are restated here: base anchors from
``octave_base_scale * 2**(i / scales_per_octave)`` scales x ratios (h = sqrt(r), w = 1 / sqrt(r);
ratio-major, scale-minor), shifted over the grid position-major / anchor-minor, centre offset 0,
and ``valid_flags`` from the padded image shape.  Checked against the reference class run on a
stand-in base class in tests/golden/make_golden_heads.py.
"""
import math

import torch


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


class RAnchorGenerator:
    """``dict(type='RAnchorGenerator', octave_base_scale=4, scales_per_octave=3, ratios=[1.0, 0.5, 2.0],
    strides=[8, 16, 32, 64, 128])`` -> per level (H*W*A, 5) rows ``(cx, cy, w, h, 0)`` (:11-39)."""

    def __init__(self, strides, ratios, scales=None, base_sizes=None, scale_major=True,
                 octave_base_scale=None, scales_per_octave=None, centers=None, center_offset=0.):
        if center_offset != 0 or centers is not None or not scale_major:
            raise NotImplementedError('only the configuration of the shipped rotated configs is restated')
        assert (scales is None) != (octave_base_scale is None and scales_per_octave is None)
        self.strides = [_pair(s) for s in strides]
        self.base_sizes = [min(s) for s in self.strides] if base_sizes is None else list(base_sizes)
        if scales is not None:
            self.scales = torch.tensor(scales, dtype=torch.float32)
        else:
            self.scales = torch.tensor([octave_base_scale * 2 ** (i / scales_per_octave)
                                        for i in range(scales_per_octave)], dtype=torch.float32)
        self.ratios = torch.tensor(ratios, dtype=torch.float32)
        self._cache = {}

    @property
    def num_levels(self):
        return len(self.strides)

    @property
    def num_base_anchors(self):
        return [self.ratios.numel() * self.scales.numel() for _ in self.strides]

    def base_wh(self, level_idx):
        """(A,), (A,): widths and heights of the level's base anchors, ratio-major."""
        h_r = self.ratios.sqrt()
        w_r = 1 / h_r
        b = self.base_sizes[level_idx]
        ws = (b * w_r[:, None] * self.scales[None, :]).reshape(-1)
        hs = (b * h_r[:, None] * self.scales[None, :]).reshape(-1)
        # mmdet builds corner boxes x -/+ 0.5 w and the rotated subclass takes their difference (:34-35)
        return (0.5 * ws) - (-0.5 * ws), (0.5 * hs) - (-0.5 * hs)

    def single_level_grid_priors(self, featmap_size, level_idx, dtype=torch.float32, device='cuda'):
        H, W = featmap_size
        sw, sh = self.strides[level_idx]
        ws, hs = self.base_wh(level_idx)
        A = ws.numel()
        ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32) * sh,
                                torch.arange(W, dtype=torch.float32) * sw, indexing='ij')
        a = torch.zeros(H * W, A, 5, dtype=torch.float32)
        # corners (shift -/+ half size) averaged back to the centre, as (x1 + x2) / 2 does (:34)
        a[:, :, 0] = ((xs.reshape(-1, 1) - 0.5 * ws) + (xs.reshape(-1, 1) + 0.5 * ws)) / 2
        a[:, :, 1] = ((ys.reshape(-1, 1) - 0.5 * hs) + (ys.reshape(-1, 1) + 0.5 * hs)) / 2
        a[:, :, 2] = (xs.reshape(-1, 1) + 0.5 * ws) - (xs.reshape(-1, 1) - 0.5 * ws)
        a[:, :, 3] = (ys.reshape(-1, 1) + 0.5 * hs) - (ys.reshape(-1, 1) - 0.5 * hs)
        return a.reshape(-1, 5).to(device=device, dtype=dtype)

    def grid_priors(self, featmap_sizes, device='cuda'):
        assert len(featmap_sizes) <= self.num_levels  # (a head used on the first levels only takes a prefix)
        key = (tuple(tuple(int(v) for v in fs) for fs in featmap_sizes), str(device))
        if key not in self._cache:
            self._cache[key] = [self.single_level_grid_priors(fs, i, device=device)
                                for i, fs in enumerate(featmap_sizes)]
        return self._cache[key]

    grid_anchors = grid_priors

    def valid_flags(self, featmap_sizes, pad_shape, device='cuda'):
        """Per level (H*W*A,) bool: positions whose cell starts inside the padded image."""
        flags = []
        for i, (fh, fw) in enumerate(featmap_sizes):
            sw, sh = self.strides[i]
            h, w = pad_shape[:2]
            vh = min(int(math.ceil(h / sh)), fh)
            vw = min(int(math.ceil(w / sw)), fw)
            vy = torch.zeros(fh, dtype=torch.bool, device=device)
            vx = torch.zeros(fw, dtype=torch.bool, device=device)
            vy[:vh] = True
            vx[:vw] = True
            v = (vy[:, None] & vx[None, :]).reshape(-1)
            A = self.num_base_anchors[i]
            flags.append(v[:, None].expand(-1, A).reshape(-1))
        return flags

    def __repr__(self):
        return (f'{self.__class__.__name__}(strides={self.strides}, ratios={self.ratios.tolist()}, '
                f'scales={self.scales.tolist()})')


class PseudoAnchorGenerator(RAnchorGenerator):
    """Valid flags only (:42-62): the refine heads use the previous stage's boxes as anchors."""

    def __init__(self, strides):
        self.strides = [_pair(s) for s in strides]

    @property
    def num_base_anchors(self):
        return [1 for _ in self.strides]

    def single_level_grid_priors(self, *args, **kwargs):
        raise NotImplementedError

    single_level_grid_anchors = single_level_grid_priors
    grid_priors = single_level_grid_priors
    grid_anchors = single_level_grid_priors

    def __repr__(self):
        return f'{self.__class__.__name__}(\n    strides={self.strides})'
