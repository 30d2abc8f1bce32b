"""Historical name of `siglip_lit_coco.py` (README.md:83, README_lit.md:38-42, BASELINE.json)."""
from big_vision_amd.configs.proj.image_text.siglip_lit_coco import get_config  # noqa: F401
