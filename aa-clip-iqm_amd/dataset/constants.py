"""Prompt/class-name tables consumed by get_adapted_text_embedding.  The values
are data exported from the reference's dataset/constants.py by
tests/golden/make_golden.py into constants.json (no code is shared).

DATA_PATH: the reference hard-codes its author's directories (dataset/constants.py:2-14); here
every dataset lives under $AACLIP_DATA_ROOT (default ./data) in a directory of its own name,
and single entries can be redirected with AACLIP_DATA_<NAME>."""
import json
import os

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "constants.json")) as _f:
    _c = json.load(_f)

CLASS_NAMES = _c["CLASS_NAMES"]
REAL_NAMES = _c["REAL_NAMES"]
DOMAINS = _c["DOMAINS"]
PROMPTS = _c["PROMPTS"]

BASE_PATH = os.environ.get("AACLIP_DATA_ROOT", "./data")
DATA_PATH = {name: os.environ.get(f"AACLIP_DATA_{name.upper()}", os.path.join(BASE_PATH, name))
             for name in CLASS_NAMES}
