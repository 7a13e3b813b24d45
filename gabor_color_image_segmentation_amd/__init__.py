"""MI355X-native Gabor-bank colour segmentation: a drop-in for the segmenter slot at
/root/reference/BSD_metrics/script.py:30 (``labels = segment(img)``)."""
from .bank import GaborBank, make_bank, gabor_taps, split_digits  # noqa: F401
from .segmenter import Segmenter, segment, segment_batch, segment_images, HipOps, lloyd, shard_rows, halo_rows  # noqa: F401
from ._lib import GcsError  # noqa: F401
