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
    "stem6": (512, 512, 6, 128, 7, 1, 3, 1, False, True),
    "rb1024_320": (64, 40, 1024, 1024, 3, 1, 1, 1, False, True),
    "rb1024_680": (64, 85, 1024, 1024, 3, 1, 1, 1, False, True),
    "head3": (512, 512, 128, 3, 7, 1, 3, 1, False, False),
    # the stride-2 / transposed layers at the reference's own geometries (fadg0: 512x320 cropped, 512x680 full width)
    "down128_320": (512, 320, 128, 256, 3, 2, 1, 0, False, True),
    "down256_320": (256, 160, 256, 512, 3, 2, 1, 0, False, True),
    "down512_320": (128, 80, 512, 1024, 3, 2, 1, 0, False, True),
    "up1024_320": (64, 40, 1024, 512, 3, 2, 1, 0, True, True),
    "up512_320": (128, 80, 512, 256, 3, 2, 1, 0, True, True),
    "up256_320": (256, 160, 256, 128, 3, 2, 1, 0, True, True),
    "down128_680": (512, 680, 128, 256, 3, 2, 1, 0, False, True),
    "down256_680": (256, 340, 256, 512, 3, 2, 1, 0, False, True),
    "down512_680": (128, 170, 512, 1024, 3, 2, 1, 0, False, True),
    "up1024_680": (64, 85, 1024, 512, 3, 2, 1, 0, True, True),
    "up512_680": (128, 170, 512, 256, 3, 2, 1, 0, True, True),
    "up256_680": (256, 340, 256, 128, 3, 2, 1, 0, True, True),
    # local enhancer of the two-scale 1024x1024 generator (ngf 64)
    "l_stem": (1024, 1024, 9, 64, 7, 1, 3, 1, False, True),
    "l_down": (1024, 1024, 64, 128, 3, 2, 1, 0, False, True),
    "l_rb": (512, 512, 128, 128, 3, 1, 1, 1, False, True),
    "l_up": (512, 512, 128, 64, 3, 2, 1, 0, True, True),
    "l_head": (1024, 1024, 64, 3, 7, 1, 3, 1, False, False),
    # the 2-scale PatchGAN discriminator of the train step (ndf 64, 4x4 convs, pad 2): scale 0 on 512x512, scale 1 on 256x256
    "d0": (512, 512, 6, 64, 4, 2, 2, 0, False, False),
    "d1": (257, 257, 64, 128, 4, 2, 2, 0, False, True),
    "d2": (129, 129, 128, 256, 4, 2, 2, 0, False, True),
    "d3": (65, 65, 256, 512, 4, 1, 2, 0, False, True),
    "d0h": (256, 256, 6, 64, 4, 2, 2, 0, False, False),
    "d1h": (129, 129, 64, 128, 4, 2, 2, 0, False, True),
    "d2h": (65, 65, 128, 256, 4, 2, 2, 0, False, True),
    "d3h": (33, 33, 256, 512, 4, 1, 2, 0, False, True),
    # the tiny maps of the train-step parity test (32x32 frames, ngf 32): few tiles, 64x64 tile config, M < BM
    "t_rb128": (8, 8, 128, 128, 3, 1, 1, 1, False, True),
    "t_down32": (32, 32, 32, 64, 3, 2, 1, 0, False, True),
    "t_down64": (16, 16, 64, 128, 3, 2, 1, 0, False, True),
    "t_up128": (8, 8, 128, 64, 3, 2, 1, 0, True, True),
    "t_up64": (16, 16, 64, 32, 3, 2, 1, 0, True, True),
    "t_dgrad_up": (16, 16, 32, 64, 3, 2, 1, 0, False, False),     # data gradient of t_up64: a stride-2 conv
    "t_dgrad_down": (8, 8, 128, 64, 3, 2, 1, 0, True, False),     # data gradient of t_down64: a transposed conv
    "t_stem": (32, 32, 9, 32, 7, 1, 3, 1, False, True),
    "t_d4x4": (32, 32, 6, 16, 4, 2, 2, 0, False, False),
}
