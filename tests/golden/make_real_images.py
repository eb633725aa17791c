#!/opt/conda/bin/python3.9
"""Step 1 of 2 of the REAL-PHOTOGRAPH fixture set (tests/golden/real/): the image files.

TUM fr3/walking_xyz (BASELINE configs 1-3) is on no box this project runs on.  The build container's conda interpreter does hold
real photographs -- scikit-image 0.18.3's data directory and scipy.misc's face / ascent: camera optics, demosaicing, JPEG blocks,
specular edges, a rectified stereo pair -- and this script turns them into the frames the extractor would be handed:

  vga/<name>.png      640 x 480.  Colour ones stay 3-channel (the PNG holds R,G,B; the loader reverses the channels into
                      cv::imread's B,G,R memory order, so the gray conversion of src/Tracking.cc:339-353 is part of the tested
                      path with both Camera.RGB settings); single-channel sources stay single-channel (cv::imread(..., UNCHANGED)
                      as the reference's RGB-D examples read them: mImGray is used as it is).
                      How each was made is in MANIFEST below: crops at native resolution where the source is large enough
                      (face_crop, motorcycle_*, hubble_crop, retina_crop), otherwise an aspect-preserving Lanczos resample to
                      cover 640 x 480 followed by a centre crop.
  native/<name>.png   native size, gray (odd sizes: 512 x 512, 600 x 400, 451 x 300, 741 x 500, 1024 x 768, 448 x 172 ...).
  jpeg/<name>_q<Q>.jpg  two of the VGA colour frames re-encoded at quality 75 and 90 (8 x 8 block artefacts, chroma subsampling).
                      JPEG decoders may differ in their IDCT: step 2 records the sha256 of the DECODED pixels, and the tests check
                      it before using them.

Run:  /opt/conda/bin/python3.9 tests/golden/make_real_images.py      (then step 2: python tests/golden/make_real_golden.py)
Sources and their licences: tests/golden/real/README.md.
"""
import os
import warnings

import numpy as np
from PIL import Image

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "real")
SK = "/opt/conda/lib/python3.9/site-packages/skimage/data/"
W, H = 640, 480


def cover_crop(im, w=W, h=H):
    """aspect-preserving Lanczos resample so that the image covers w x h, then centre crop"""
    s = max(w / im.width, h / im.height)
    nw, nh = max(w, int(round(im.width * s))), max(h, int(round(im.height * s)))
    im = im.resize((nw, nh), Image.LANCZOS)
    x0, y0 = (nw - w) // 2, (nh - h) // 2
    return im.crop((x0, y0, x0 + w, y0 + h))


def crop(im, x0, y0, w=W, h=H):
    assert x0 + w <= im.width and y0 + h <= im.height
    return im.crop((x0, y0, x0 + w, y0 + h))


def main():
    import scipy.misc
    for sub in ("vga", "native", "jpeg"):
        os.makedirs(os.path.join(OUT, sub), exist_ok=True)
    sk = lambda f: Image.open(SK + f)
    face = Image.fromarray(scipy.misc.face())                          # 1024 x 768 RGB raccoon
    ascent = Image.fromarray(scipy.misc.ascent().astype(np.uint8))     # 512 x 512 gray
    vga = {
        # colour, native-resolution crops
        "face_crop": crop(face, 300, 150),
        "motorcycle_left": crop(sk("motorcycle_left.png").convert("RGB"), 60, 10),
        "motorcycle_right": crop(sk("motorcycle_right.png").convert("RGB"), 60, 10),   # same window: rows stay rectified
        "hubble_crop": crop(sk("hubble_deep_field.jpg").convert("RGB"), 180, 200),
        "retina_crop": crop(sk("retina.jpg").convert("RGB"), 400, 450),
        # colour, resampled
        "face_small": face.resize((W, H), Image.LANCZOS),               # 1024 x 768 is 4:3 already
        "astronaut": cover_crop(sk("astronaut.png").convert("RGB")),
        "coffee": cover_crop(sk("coffee.png").convert("RGB")),
        "chelsea": cover_crop(sk("chelsea.png").convert("RGB")),
        "rocket": cover_crop(sk("rocket.jpg").convert("RGB")),
        # single channel
        "camera": cover_crop(sk("camera.png").convert("L")),
        "brick": cover_crop(sk("brick.png").convert("L")),
        "grass": cover_crop(sk("grass.png").convert("L")),
        "gravel": cover_crop(sk("gravel.png").convert("L")),
        "page": cover_crop(sk("page.png").convert("L")),
        "text": cover_crop(sk("text.png").convert("L")),
        "coins": cover_crop(sk("coins.png").convert("L")),
        "ascent": cover_crop(ascent),
    }
    for name, im in vga.items():
        assert im.size == (W, H), (name, im.size)
        im.save(os.path.join(OUT, "vga", name + ".png"), optimize=True)
    native = {
        "camera": sk("camera.png").convert("L"),               # 512 x 512
        "coffee": sk("coffee.png").convert("L"),               # 600 x 400
        "chelsea": sk("chelsea.png").convert("L"),             # 451 x 300
        "rocket": sk("rocket.jpg").convert("L"),               # 640 x 427, a camera JPEG as shipped
        "motorcycle_left": sk("motorcycle_left.png").convert("L"),    # 741 x 500
        "face": face.convert("L"),                             # 1024 x 768
        "text": sk("text.png").convert("L"),                   # 448 x 172: only 4 pyramid levels fit
        "page": sk("page.png").convert("L"),                   # 384 x 191
    }
    for name, im in native.items():
        im.save(os.path.join(OUT, "native", name + ".png"), optimize=True)
    for name in ("face_crop", "coffee"):
        for q in (75, 90):
            vga[name].save(os.path.join(OUT, "jpeg", f"{name}_q{q}.jpg"), quality=q)
    total = sum(os.path.getsize(os.path.join(r, f)) for r, _, fs in os.walk(OUT) for f in fs)
    print(f"{len(vga)} vga + {len(native)} native + 4 jpeg files, {total / 1e6:.2f} MB under {OUT}")


if __name__ == "__main__":
    main()
