"""print_ppm (lib.rs:344-361): sqrt gamma, `(255.99 * x) as i32` clamped to 0..=255, ASCII P3.
Host post-process of the framebuffer the hot path returns."""
import numpy as np


def to_u8(img):
    col = np.sqrt(np.asarray(img, dtype=np.float32))
    v = np.float32(255.99) * col
    with np.errstate(invalid="ignore"):
        i = np.where(np.isnan(v), 0, np.clip(v, -2147483648.0, 2147483647.0)).astype(np.int64)
    return np.clip(i, 0, 255).astype(np.int32)


def format_ppm(img):
    ny, nx, _ = img.shape
    px = to_u8(img).reshape(-1, 3)
    return "P3\n%d %d\n255\n" % (nx, ny) + "".join("%d %d %d\n" % (r, g, b) for r, g, b in px)
