"""Prompt/class-name tables consumed by get_adapted_text_embedding.  The values
are data exported from the reference's dataset/constants.py by
tests/golden/make_golden.py into constants.json (no code is shared)."""
import json
import os

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "constants.json")) as _f:
    _c = json.load(_f)

CLASS_NAMES = _c["CLASS_NAMES"]
REAL_NAMES = _c["REAL_NAMES"]
DOMAINS = _c["DOMAINS"]
PROMPTS = _c["PROMPTS"]
