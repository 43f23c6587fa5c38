"""Natural-content fixtures from the pictures the reference checkout holds (build container only: needs /root/reference and PIL).

    python tests/golden/make_natural.py          ->  tests/golden/natural_luma.npz

What it records (data only: uint8 luminance planes, no reference source text):
  cactus, kimono, parkscene   768 x 1152 windows of the first-frame luminance PNGs of three HEVC class-B sequences
                              (hevc/visualization/map_intra_prediction_modes/readme/luminance_{cactus,kimono,parkscene}.png)
  cliff, library              luminance of hevc/pseudo_data/rgb_{cliff,library}.jpg, converted by the REFERENCE's own
                              tools.tools.rgb_to_ycbcr (imported from /root/reference), cropped to 640 x 960
Provenance: the three PNGs are first frames of JCT-VC class-B test sequences (Cactus, Kimono, ParkScene) as the reference's authors
published them in their repository; the two JPEGs are the authors' own test photographs.  The fixture is therefore GENERATED where the
reference checkout exists (__graft_entry__.build() calls this script) and git-ignored: it travels to the GPU box with the working tree
and is never committed or redistributed; tests and campaigns that need it skip / refuse when it is absent.
The trained networks were trained on natural (ImageNet) luminance: these are the only natural pictures the repository holds, and
what the parity / evidence tests (tests/test_natural.py) and the natural-picture HM campaigns (tools/hm/campaign.py) run on.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def main():
    from PIL import Image
    sys.path.insert(0, REF)
    if not hasattr(np, "float"):
        np.float = float                               # the reference predates numpy 1.24 (numpy.float alias)
    import tools.tools as tls                          # the reference's own colour conversion
    out = {}
    for name, (y0, x0) in (("cactus", (156, 384)), ("kimono", (156, 384)), ("parkscene", (156, 384))):
        a = np.asarray(Image.open(os.path.join(REF, "hevc/visualization/map_intra_prediction_modes/readme/luminance_%s.png" % name)))
        assert a.dtype == np.uint8 and a.shape == (1080, 1920)
        out[name] = np.ascontiguousarray(a[y0:y0 + 768, x0:x0 + 1152])
    for name in ("cliff", "library"):
        rgb = np.asarray(Image.open(os.path.join(REF, "hevc/pseudo_data/rgb_%s.jpg" % name)))
        assert rgb.dtype == np.uint8 and rgb.shape == (641, 960, 3)
        out[name] = np.ascontiguousarray(tls.rgb_to_ycbcr(rgb)[:640, :, 0])
    path = os.path.join(HERE, "natural_luma.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), {k: (v.shape, float(v.mean())) for k, v in out.items()})


if __name__ == "__main__":
    main()
