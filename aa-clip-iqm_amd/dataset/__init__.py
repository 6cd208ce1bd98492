"""Only the constants that feed the text-anchor path are provided (prompts, class
names, domains: reference dataset/constants.py:16-148).  Image loading is host
I/O outside the hot path."""
