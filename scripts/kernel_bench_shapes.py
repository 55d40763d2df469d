"""Layer shapes of the 512x512 generator (shared by the micro-benchmarks)."""
SHAPES = {
    # name: (H, W, Cin, Cout, k, stride, pad, pad_mode, transposed, stats)
    "rb1024": (64, 64, 1024, 1024, 3, 1, 1, 1, False, True),
    "down512": (128, 128, 512, 1024, 3, 2, 1, 0, False, True),
    "down256": (256, 256, 256, 512, 3, 2, 1, 0, False, True),
    "down128": (512, 512, 128, 256, 3, 2, 1, 0, False, True),
    "up1024": (64, 64, 1024, 512, 3, 2, 1, 0, True, True),
    "up512": (128, 128, 512, 256, 3, 2, 1, 0, True, True),
    "up256": (256, 256, 256, 128, 3, 2, 1, 0, True, True),
    "stem9": (512, 512, 9, 128, 7, 1, 3, 1, False, True),
    "rb1024_320": (64, 40, 1024, 1024, 3, 1, 1, 1, False, True),
    "rb1024_680": (64, 85, 1024, 1024, 3, 1, 1, 1, False, True),
    "head3": (512, 512, 128, 3, 7, 1, 3, 1, False, False),
}
