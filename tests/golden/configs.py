"""Reduced-width configurations shared by the golden generator and the tests (same constructor kwargs as the
shipped YAMLs, configs/mgldvsr/mgldvsr_512_realbasicvsr_deg.yaml:34-52,87-108 and
configs/video_vae/video_autoencoder_kl_64x64x4_resi.yaml:36-52, only narrower)."""

T = 3

UNET_SMALL = dict(
    num_frames=T, image_size=32, in_channels=4, out_channels=4, model_channels=64, attention_resolutions=[4, 2, 1],
    num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_head_channels=64, use_spatial_transformer=True,
    use_linear_in_transformer=True, transformer_depth=1, context_dim=64, use_checkpoint=False, legacy=False,
    semb_channels=64)

STRUCT_SMALL = dict(
    num_frames=T, image_size=96, in_channels=4, model_channels=64, out_channels=64, num_res_blocks=2,
    attention_resolutions=[4, 2, 1], dropout=0, channel_mult=[1, 1, 2, 2], conv_resample=True, dims=2,
    use_checkpoint=False, use_fp16=False, num_heads=1, num_head_channels=-1, num_heads_upsample=-1,
    use_scale_shift_norm=False, resblock_updown=False, use_new_attention_order=False)

VAE_DD_SMALL = dict(double_z=True, num_frames=T, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=32,
                    ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)

# full-size (shipped) configurations
UNET_FULL = dict(
    num_frames=5, image_size=32, in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1],
    num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_head_channels=64, use_spatial_transformer=True,
    use_linear_in_transformer=True, transformer_depth=1, context_dim=1024, use_checkpoint=False, legacy=False,
    semb_channels=256)

STRUCT_FULL = dict(
    num_frames=5, image_size=96, in_channels=4, model_channels=256, out_channels=256, num_res_blocks=2,
    attention_resolutions=[4, 2, 1], dropout=0, channel_mult=[1, 1, 2, 2], conv_resample=True, dims=2,
    use_checkpoint=False, use_fp16=False, num_heads=4, num_head_channels=-1, num_heads_upsample=-1,
    use_scale_shift_norm=False, resblock_updown=False, use_new_attention_order=False)

VAE_DD_FULL = dict(double_z=True, num_frames=5, z_channels=4, resolution=512, in_channels=3, out_ch=3, ch=128,
                   ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)
