"""Small-network hyper-parameters shared by tests/golden/make_golden.py's fixtures and the tests that replay them
(real topology of the released UNet / AE, narrow widths; widths are multiples of 64 = one MFMA K slice)."""
TINY_UNET = dict(in_channels=8, out_channels=4, model_channels=64, attention_resolutions=[4, 2, 1], num_res_blocks=2,
                 channel_mult=[1, 2, 4, 4], dropout=0.1, num_head_channels=64, transformer_depth=1, context_dim=128,
                 use_linear=True, use_checkpoint=False, temporal_conv=True, temporal_attention=True,
                 temporal_selfatt_only=True, use_relative_position=False, use_causal_attention=False,
                 temporal_length=4, addition_attention=True, image_cross_attention=True, default_fs=10,
                 fs_condition=True)
TINY_AE = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=64, ch_mult=[1, 2, 4, 4],
               num_res_blocks=2, attn_resolutions=[], dropout=0.0)
