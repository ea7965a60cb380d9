"""Tiny stand-in for the two OpenCV calls the reference's harness makes (examples/run_test.py:40-41):
cvtColor(img, COLOR_RGB2BGR) and imwrite("*.exr" | "*.npy", img).  Only used when the real cv2 is
not installed (this directory sits behind site-packages on sys.path only if the caller puts it there)."""
import numpy as np

COLOR_RGB2BGR = 4
COLOR_BGR2RGB = 4


def cvtColor(img, code):
    return np.ascontiguousarray(np.asarray(img)[..., ::-1])


def imwrite(path, img):
    img = np.asarray(img, dtype=np.float32)
    if path.lower().endswith(".exr"):
        from psdr_cuda.exr import save_exr_rgb
        save_exr_rgb(path, img[..., ::-1])          # OpenCV images are BGR
    else:
        np.save(path if path.endswith(".npy") else path + ".npy", img)
    return True
