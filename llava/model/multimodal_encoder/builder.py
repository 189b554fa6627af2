from halva_amd.clip import build_vision_tower  # noqa: F401
