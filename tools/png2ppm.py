#!/usr/bin/env python3
"""PNG -> binary PPM (P6) and PGM/PPM -> PNG helper for examples/adcensus_cli (PIL; OpenCV is not needed)."""
import sys
from PIL import Image
src, dst = sys.argv[1], sys.argv[2]
img = Image.open(src)
if dst.lower().endswith(".ppm"):
    img = img.convert("RGB")
img.save(dst)
