"""Kernel-name prefixes of the MFMA kernel families bench.py's `roofline.families` reports (ops.ConvProfile family names), shared by
tools/pmc_busy.py and tools/pmc_traffic.py so that every family of the driver's line has its counters in the committed PMC records."""
FAMILIES = {
    # the dense-block launches: conv_chain_kernel (fp32 matrix core) or conv_sweep4_kernel / conv_sweep_kernel (bf16x3, use_amp)
    "conv_chain": ("conv_chain_kernel", "conv_sweep_kernel", "conv_sweep4_kernel"),
    "conv_tile_3x3": ("conv_tile_kernel<0,", "conv3x3_x3w8_kernel", "conv3x3_d4_kernel"),
    "conv_wino_3x3": ("conv3x3_wino_kernel",),          # the Winograd F(2x2, 3x3) form of the >= 128-channel 3x3 layers (csrc/conv_wino.hip)
    "conv_tile_3x3_up2": ("conv_tile_kernel<1,",),
    "conv_tile_4x4s2": ("conv_tile_kernel<2,", "conv_s2_d4_kernel<2,"),
    "conv_tile_dgrad4x4s2": ("conv_tile_kernel<3,", "conv_s2_d4_kernel<3,"),
    "conv_tile_1x1": ("conv_tile_kernel<4,",),
    "conv_tile_3x3_c4": ("conv_tile_kernel<5,",),
    "conv_tile_7x7_c4": ("conv_tile_kernel<6,",),       # config 5: ResnetGenerator's 7x7 image-side layers, 49 taps in K
    "wgrad_tile": ("wgrad_tile_kernel",),
}
