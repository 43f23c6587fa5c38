"""Batched L-shaped context extraction on the GPU, mirroring the reference's numpy path.

Mirrors sets/common.py:265-349 `extract_context_portions_targets_from_channels_plus_preprocessing`
(-> :13-110 slicing, :351-475 mean subtraction, rectangular masks, FC flatten-concat): same arguments, same
return tuples, same exceptions for bad arguments.  The portions are produced by the HIP gather kernel
(uint8 planes, rectangular masks expressed as unit masks); targets are sliced on the host (they are not
on the prediction path)."""
import ctypes

import numpy as np

from . import _lib


def extract_context_portions_targets_from_channels_plus_preprocessing(channels_single_or_pair_uint8, width_target, row_1sts,
                                                                      col_1sts, mean_training, tuple_width_height_masks,
                                                                      is_fully_connected, predictor=None):
    """Returns (flattened_contexts [N,5w^2], targets [N,w,w,1]) or (above [N,w,3w,1], left [N,2w,w,1], targets).

    `predictor` supplies the libpnn_hip context (any PredictionNeuralNetwork on the target device)."""
    import torch
    ch = channels_single_or_pair_uint8
    if ch.dtype != np.uint8:
        raise TypeError('`channels_single_or_pair_uint8.dtype` is not equal to `numpy.uint8`.')
    if not np.issubdtype(row_1sts.dtype, np.integer):
        raise TypeError('`row_1sts.dtype` is not smaller than `numpy.integer` in type hierarchy.')
    if not np.issubdtype(col_1sts.dtype, np.integer):
        raise TypeError('`col_1sts.dtype` is not smaller than `numpy.integer` in type hierarchy.')
    if col_1sts.size != row_1sts.size:
        raise ValueError('`col_1sts.size` is not equal to `row_1sts.size`.')
    nb_images, height, width, nb_channels = ch.shape
    if nb_channels not in (1, 2):
        raise ValueError('`channel_single_or_pair_uint8.shape[2]` does not belong to {1, 2}.')
    w = width_target
    (mask_w, mask_h) = tuple_width_height_masks
    if mask_w < 0 or mask_w > w or mask_w % 4 != 0:
        raise ValueError('`tuple_width_height_masks[0]` does not belong to {0, 4, ..., `targets_uint8.shape[1]`}.')
    if mask_h < 0 or mask_h > w or mask_h % 4 != 0:
        raise ValueError('`tuple_width_height_masks[1]` does not belong to {0, 4, ..., `targets_uint8.shape[1]`}.')
    for r, c in zip(row_1sts.tolist(), col_1sts.tolist()):
        if r < 0 or c < 0:
            raise ValueError('`row_1st` / `col_1st` is not positive.')
        if r + 3 * w > height or c + 3 * w > width:
            raise ValueError('the context does not fit into the channel.')
    if predictor is None:
        raise ValueError("`predictor` (a PredictionNeuralNetwork holding the GPU context) is required")
    L = _lib.lib()
    n_pos = row_1sts.size
    n = nb_images * n_pos
    units = 2 * w // 4
    # context portions come from the LAST channel (the HEVC-decoded one of a pair), targets from channel 0
    plane = np.ascontiguousarray(ch[:, :, :, nb_channels - 1])
    tbs = (_lib.TbDev * n)()
    above_mask = (1 << (units - mask_w // 4)) - 1
    for i in range(nb_images):
        for k in range(n_pos):
            d = tbs[i * n_pos + k]
            d.origin = (i * height + int(row_1sts[k]) + w) * width + int(col_1sts[k]) + w
            d.stride = width
            d.above_mask = above_mask
            d.left_units = units - mask_h // 4
    d_plane = torch.from_numpy(plane).cuda(predictor.device)
    d_tbs = torch.from_numpy(np.frombuffer(tbs, dtype=np.uint8).copy()).cuda(predictor.device)
    stream = ctypes.c_void_p(torch.cuda.current_stream(predictor.device).cuda_stream)
    ctx_mean = ctypes.c_float(L.pnn_mean(predictor.ctx)).value
    if abs(ctx_mean - np.float32(mean_training)) > 1e-6:
        raise ValueError("`mean_training` differs from the predictor's mean")
    if is_fully_connected:
        d_ctx = torch.empty((n, 5 * w * w), dtype=torch.float32, device=d_plane.device)
        rc = L.pnn_gather_device(predictor.ctx, w, 4, d_plane.data_ptr(), 1, d_tbs.data_ptr(), n, d_ctx.data_ptr(),
                                 5 * w * w, d_ctx.data_ptr() + 4 * 3 * w * w, 5 * w * w, stream)
        _lib.check(rc, predictor.ctx)
        outs = (d_ctx.cpu().numpy(),)
    else:
        d_above = torch.empty((n, w, 3 * w, 1), dtype=torch.float32, device=d_plane.device)
        d_left = torch.empty((n, 2 * w, w, 1), dtype=torch.float32, device=d_plane.device)
        rc = L.pnn_gather_device(predictor.ctx, w, 4, d_plane.data_ptr(), 1, d_tbs.data_ptr(), n, d_above.data_ptr(),
                                 3 * w * w, d_left.data_ptr(), 2 * w * w, stream)
        _lib.check(rc, predictor.ctx)
        outs = (d_above.cpu().numpy(), d_left.cpu().numpy())
    targets = np.zeros((n, w, w, 1), np.float32)
    for i in range(nb_images):
        for k in range(n_pos):
            r, c = int(row_1sts[k]) + w, int(col_1sts[k]) + w
            targets[i * n_pos + k] = ch[i, r:r + w, c:c + w, 0:1].astype(np.float32) - np.float32(mean_training)
    return outs + (targets,)
