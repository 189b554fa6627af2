from halva_amd.clip import build_vision_projector  # noqa: F401
