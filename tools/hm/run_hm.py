"""Encode / decode harness for the reference's modified HM-16.15 codecs built on libpnn_hip.so (tools/hm/Makefile).

What the reference's experiment scripts do around the codecs (hevc/running.py:60-135: write the 4:0:0 picture, run
`TAppEncoderStatic -c cfg -i in.yuv -b str.bin -o rec.yuv -wdt W -hgt H --InputBitDepth=8 --InputChromaFormat=400
--FramesToBeEncoded=1 --QP=q` with the three PNN options, then `TAppDecoderStatic -b str.bin -o dec.yuv` with the same
three options; hevc/performance.py:12-46: scrape `Total Time`), restated for BASELINE.json configs[3]/[4] with
synthetic pictures and seeded random-init models (the trained production weights are not in the reference checkout).

    python tools/hm/run_hm.py --variant substitution --width 256 --height 192 --qp 32 --out gpurun_out/hm
    python tools/hm/run_hm.py --variant switch --service --jobs 4 ...      # 4 concurrent encodes behind one batching service

Prints one JSON line per encode (HM Total Time, bits, PSNR, PNN calls per width, cache hits, encoder/decoder match).
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MEAN = 117.8952234192841
# hevc/configuration/intra_main_rext.cfg of the reference (all-intra, CTU 64, TU 4..32, WPP off), as key: value pairs
CFG = {
    "FrameRate": 25, "Profile": "main-RExt", "Tier": "main", "Level": 5.2,
    "MaxCUWidth": 64, "MaxCUHeight": 64, "MaxPartitionDepth": 4,
    "QuadtreeTULog2MaxSize": 5, "QuadtreeTULog2MinSize": 2, "QuadtreeTUMaxDepthInter": 3, "QuadtreeTUMaxDepthIntra": 3,
    "IntraPeriod": 1, "DecodingRefreshType": 1, "GOPSize": 1, "ReWriteParamSetsFlag": 1,
    "FastSearch": 1, "SearchRange": 64, "HadamardME": 1, "FEN": 1, "FDM": 1,
    "MaxDeltaQP": 0, "MaxCuDQPDepth": 0, "DeltaQpRD": 0, "RDOQ": 1, "RDOQTS": 1,
    "LoopFilterOffsetInPPS": 1, "LoopFilterDisable": 0, "LoopFilterBetaOffset_div2": 0, "LoopFilterTcOffset_div2": 0,
    "DeblockingFilterMetric": 0, "SAO": 1, "AMP": 1, "TransformSkip": 1, "TransformSkipFast": 1, "SAOLcuBoundary": 0,
    "SliceMode": 0, "SliceArgument": 1500, "LFCrossSliceBoundaryFlag": 1,
    "PCMEnabledFlag": 0, "PCMLog2MaxSize": 5, "PCMLog2MinSize": 3, "PCMInputBitDepthFlag": 1, "PCMFilterDisableFlag": 0,
    "TileUniformSpacing": 0, "NumTileColumnsMinus1": 0, "NumTileRowsMinus1": 0, "LFCrossTileBoundaryFlag": 1,
    "WaveFrontSynchro": 0, "ScalingList": 0, "TransquantBypassEnable": 0, "CUTransquantBypassFlagForce": 0,
    "SEIDecodedPictureHash": 3,
}


def exe(variant, which):
    p = os.path.join(HERE, "_build", variant, "TApp%sStatic" % which)
    if not os.path.exists(p):
        raise FileNotFoundError("%s is missing: run `make -C tools/hm` where /root/reference exists" % p)
    return p


def write_cfg(path):
    with open(path, "w") as f:
        for k, v in CFG.items():
            f.write("%-28s: %s\n" % (k, v))


def make_frame(h, w, seed):
    """Seeded synthetic luminance picture: 8x8 / 16x16 / 32x32 patches of constant grey with mild noise; a share of the
    patches sits at the training mean (118), where a PNN that has learnt nothing (random init, output ~ 0 + mean) is
    the best predictor -- so the encoder does select the PNN mode and the decoder has to reproduce it."""
    rng = np.random.RandomState(seed)
    img = np.zeros((h, w), np.float32)
    for size, y0, y1 in ((8, 0, h // 3), (16, h // 3, 2 * h // 3), (32, 2 * h // 3, h)):
        ny, nx = (y1 - y0 + size - 1) // size, (w + size - 1) // size
        lv = rng.randint(20, 236, (ny, nx)).astype(np.float32)
        lv[rng.rand(ny, nx) < 0.45] = 118.0
        img[y0:y1] = np.kron(lv, np.ones((size, size), np.float32))[:y1 - y0, :w]
    img += rng.normal(0, 1.5, (h, w))
    return np.clip(np.round(img), 0, 255).astype(np.uint8)


def natural_frames(h, w, n, seed=1):
    """n uint8 luminance pictures of h x w cut from the natural fixtures (tests/golden/natural_luma.npz: windows of three HEVC class-B
    first frames and two photographs of the reference checkout, tests/golden/make_natural.py): windows on a 64-pixel grid, every second one
    mirrored -- distinct pictures with natural statistics, since Kodak / BSDS themselves are not in the checkout."""
    pics = np.load(os.path.join(ROOT, "tests", "golden", "natural_luma.npz"))
    names = sorted(pics.files)
    rng = np.random.RandomState(seed)
    out, seen = [], set()
    while len(out) < n:
        k = len(out) % len(names)
        img = pics[names[k]]
        H, W = img.shape
        y = 64 * rng.randint(0, (H - h) // 64 + 1) if H > h else 0
        x = 64 * rng.randint(0, (W - w) // 64 + 1) if W > w else 0
        flip = (len(out) // len(names)) % 2 == 1
        key = (k, y, x, flip)
        if key in seen and len(seen) < 200:
            continue
        seen.add(key)
        win = img[y:y + h, x:x + w]
        out.append(np.ascontiguousarray(win[:, ::-1] if flip else win))
    return out


def model_params(w, seed=11, trained_small=False):
    """(flat parameters, is_fc) of the model make_models writes for width w."""
    from context_adaptive_neural_network_based_prediction_amd import weights as wts
    if trained_small and w <= 8:
        flat, _, _ = wts.load_pnnw(os.path.join(ROOT, "tests", "golden", "conv%d_single.pnnw" % w))
        return flat, False
    return wts.init_params(w, w <= 8, seed + w, bias_std=0.02), w <= 8


def make_models(out_dir, seed=11, trained_small=False, only_widths=(4, 8, 16, 32, 64)):
    """Seeded random-init models in the reference's architecture per width (FC for 4, 8; conv for 16, 32, 64;
    PredictionNeuralNetwork.py:119-137) as .pnnw files + the `width,is_pair,channel,path` table + the mean file
    (pickle protocol 2, as sets/results/training_set/means/luminance/mean_training.pkl)."""
    import pickle

    from context_adaptive_neural_network_based_prediction_amd import weights as wts
    os.makedirs(out_dir, exist_ok=True)
    lines = []
    for w in (4, 8, 16, 32, 64):
        is_fc = w <= 8
        path = os.path.join(out_dir, "pnn_%d.pnnw" % w)
        if w not in only_widths:
            pass                                       # listed in the table, file absent (tests of the failure paths)
        elif trained_small and w <= 8:
            # the only trained weights the reference ships: convolutional 4x4 / 8x8 (tests/golden/conv{4,8}_single.pnnw)
            import shutil
            shutil.copy(os.path.join(ROOT, "tests", "golden", "conv%d_single.pnnw" % w), path)
        else:
            wts.save_pnnw(path, wts.init_params(w, is_fc, seed + w, bias_std=0.02), w, is_fc)
        lines.append("%d,0,0,%s" % (w, path))
    table = os.path.join(out_dir, "single.txt")
    with open(table, "w") as f:
        f.write("\n".join(lines) + "\n")
    mean_path = os.path.join(out_dir, "mean_training.pkl")
    with open(mean_path, "wb") as f:
        pickle.dump(MEAN, f, protocol=2)
    return table, mean_path


def psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 10.0 * np.log10(255.0 ** 2 / (mse + 1e-6))          # tools/tools.py:364-401


def parse_stats(stderr_text):
    """`[pnn] session width W (...): N Run calls, H answered from the cache` lines of pnn_tf_compat.h (PNN_STATS=1)."""
    out = {}
    for m in re.finditer(r"\[pnn\] session width (\d+) \(([^)]*)\): (\d+) Run calls, (\d+) answered from the cache", stderr_text):
        out[int(m.group(1))] = {"runs": int(m.group(3)), "cache_hits": int(m.group(4)), "kind": m.group(2)}
    return out


def encode_decode(variant, frame, qp, table, mean_path, work, tag="0", env=None, decoder_env=None, timeout=3600, chroma=None):
    """One encode + decode of `frame` (uint8 [H][W], 4:0:0; with `chroma` = (Cb, Cr) planes [H/2][W/2]: 4:2:0, where the PNN also
    predicts chroma blocks with 2-pixel availability units).  Returns a dict; raises on a codec failure."""
    os.makedirs(work, exist_ok=True)
    h, w = frame.shape
    fmt = "400" if chroma is None else "420"
    nbytes = h * w if chroma is None else h * w * 3 // 2
    base = os.path.join(work, "%s_%s_qp%d" % (variant, tag, qp))
    cfg = os.path.join(work, "intra_rext_400.cfg")
    if not os.path.exists(cfg):
        write_cfg(cfg)
    with open(base + "_in.yuv", "wb") as f:
        f.write(frame.tobytes())
        if chroma is not None:
            f.write(np.ascontiguousarray(chroma[0], np.uint8).tobytes())
            f.write(np.ascontiguousarray(chroma[1], np.uint8).tobytes())
    # hm_16_15_regular = the reference's stock HM-16.15 (+ mode statistics): no PNN, no extra options; the CPU-only yardstick
    pnn_args = [] if variant == "regular" else ["--PathToAdditionalDirectory=%s" % work, "--PathToMeanTraining=%s" % mean_path,
                                                "--PathToFilePathsToGraphsOutput=%s" % table]
    e = dict(os.environ)
    e["PNN_STATS"] = "1"
    e.update(env or {})
    t0 = time.time()
    enc = subprocess.run([exe(variant, "Encoder"), "-c", cfg, "-i", base + "_in.yuv", "-b", base + ".bin", "-o", base + "_rec.yuv",
                          "-wdt", str(w), "-hgt", str(h), "--InputBitDepth=8", "--InputChromaFormat=%s" % fmt, "--FramesToBeEncoded=1",
                          "--QP=%d" % qp] + pnn_args, env=e, capture_output=True, text=True, timeout=timeout)
    t_enc = time.time() - t0
    if enc.returncode != 0:
        raise RuntimeError("encoder failed (%d):\n%s\n%s" % (enc.returncode, enc.stdout[-2000:], enc.stderr[-2000:]))
    t0 = time.time()
    if decoder_env is not None:                        # e.g. encoder behind the batching service, decoder on its own context
        e = dict(os.environ)
        e["PNN_STATS"] = "1"
        e.update(decoder_env)
    dec = subprocess.run([exe(variant, "Decoder"), "-b", base + ".bin", "-o", base + "_dec.yuv", "-d", "8"] + pnn_args, env=e,
                         capture_output=True, text=True, timeout=timeout)
    t_dec = time.time() - t0
    if dec.returncode != 0:
        raise RuntimeError("decoder failed (%d):\n%s\n%s" % (dec.returncode, dec.stdout[-2000:], dec.stderr[-2000:]))
    rec_all = np.fromfile(base + "_rec.yuv", np.uint8)[:nbytes]
    dcd_all = np.fromfile(base + "_dec.yuv", np.uint8)[:nbytes]
    rec = rec_all[:h * w].reshape(h, w)
    m = re.search(r"Total Time:\s*([0-9.]+)\s*sec", enc.stdout)                 # hevc/performance.py:33
    md = re.search(r"Total Time:\s*([0-9.]+)\s*sec", dec.stdout)
    md5_bad = "ERROR" in dec.stdout or "***ERROR***" in dec.stdout               # SEIDecodedPictureHash mismatch report
    return {
        "variant": variant, "qp": qp, "width": w, "height": h,
        "enc_total_time_s": float(m.group(1)) if m else None, "enc_wall_s": round(t_enc, 3),
        "dec_total_time_s": float(md.group(1)) if md else None, "dec_wall_s": round(t_dec, 3),
        "bits": 8 * os.path.getsize(base + ".bin"),
        "psnr_rec_db": round(psnr(frame, rec), 3),
        "chroma_format": fmt,
        "decoder_equals_encoder": bool(rec_all.size == nbytes and np.array_equal(rec_all, dcd_all)), "decoder_hash_error": md5_bad,
        "enc_pnn": parse_stats(enc.stderr), "dec_pnn": parse_stats(dec.stderr),
    }


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--variant", default="substitution", choices=["substitution", "switch", "regular"])
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--qp", type=int, default=32)
    ap.add_argument("--jobs", type=int, default=1, help="concurrent encodes (different seeds)")
    ap.add_argument("--service", action="store_true", help="serve all encodes from batching service processes (one per entry of --devices)")
    ap.add_argument("--window-us", type=int, default=0, help="batching window of the service(s); 0 measured fastest (DESIGN.md 5b)")
    ap.add_argument("--devices", default="0", help="comma-separated HIP devices, one service each; encodes are dealt round-robin "
                                                     "(the multi-GPU form of the HM path: replicas, no collective)")
    ap.add_argument("--trained-small", action="store_true", help="widths 4 / 8 use the trained conv checkpoints")
    ap.add_argument("--yuv420", action="store_true", help="4:2:0 pictures (the PNN then also predicts chroma blocks, 2-pixel availability units)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "hm"))
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    table, mean_path = (None, None) if args.variant == "regular" else make_models(os.path.join(args.out, "models"), trained_small=args.trained_small)
    devices = [int(d) for d in args.devices.split(",")]
    servers, socks = [], []
    if args.service:
        for k, dev in enumerate(devices):
            sock = os.path.join(args.out, "pnn%d.sock" % k)
            servers.append(subprocess.Popen([sys.executable, "-m", "context_adaptive_neural_network_based_prediction_amd.service", "--socket", sock,
                                             "--table", table, "--device", str(dev), "--max-batch", "64", "--window-us", str(args.window_us)], cwd=ROOT,
                                            stdout=subprocess.PIPE, text=True))
            socks.append(sock)
        for srv in servers:
            line = srv.stdout.readline()
            assert "listening" in line, line

    def job_env(j):                                    # encode j -> service (or device) j % len(devices)
        k = j % len(devices)
        return {"PNN_SERVICE_SOCKET": socks[k]} if args.service else {"PNN_DEVICE": str(devices[k])}
    try:
        from concurrent.futures import ThreadPoolExecutor
        frames = [make_frame(args.height, args.width, args.seed + j) for j in range(args.jobs)]
        t0 = time.time()
        with ThreadPoolExecutor(args.jobs) as ex:
            chroma = [(make_frame(args.height // 2, args.width // 2, args.seed + 1000 + j), make_frame(args.height // 2, args.width // 2, args.seed + 2000 + j))
                      if args.yuv420 else None for j in range(args.jobs)]
            res = list(ex.map(lambda j: encode_decode(args.variant, frames[j], args.qp, table, mean_path, args.out, tag=str(j), env=job_env(j),
                                                      chroma=chroma[j]), range(args.jobs)))
        wall = time.time() - t0
        for r in res:
            print(json.dumps(r))
        print(json.dumps({"jobs": args.jobs, "service": bool(args.service), "devices": devices, "wall_s": round(wall, 2)}))
    finally:
        for srv in servers:
            srv.terminate()
        for srv in servers:
            print(srv.stdout.read().strip())
            srv.wait(10)


if __name__ == "__main__":
    main()
