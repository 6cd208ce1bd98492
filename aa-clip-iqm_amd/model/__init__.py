"""Reference-compatible import surface (`from model.clip import create_model`,
`from model.adapter import AdaptedCLIP`) backed by the gfx950 HIP path."""
