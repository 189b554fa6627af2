from halva_amd.clip import CLIPVisionTower  # noqa: F401
