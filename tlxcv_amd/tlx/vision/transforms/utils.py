"""tensorlayerx.vision.transforms.utils.load_image (demo/image_classification/predict.py:3,21)."""
import numpy as np


def load_image(path):
    """Read an image file into an HWC uint8 RGB array."""
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"))
