"""Training configurations of the reference's shipped recipes as VALUES (the YAML files themselves live in the reference repository and do
not travel): egs/ema/voc1/conf/e2w_hifigan_car.yaml ("car"), e2w_hifigan.yaml ("e2w"), egs/mri/voc1/conf/mri2w_hifigan_car.yaml ("mri") —
the keys articulatory_amd/bin/train.py::Trainer reads.  Used by bench.py's training leg, tools/gan_bench.py and the recipe-size parity tests
(tests/test_gpu_recipe.py holds a Trainer built from this against the REAL reference's Trainer._train_step on the YAML itself)."""
import copy

from .synth import disc_params

_GENERATOR = dict(
    in_channels=141, out_channels=1, channels=512, kernel_size=7, upsample_scales=[5, 4, 2, 2], upsample_kernel_sizes=[10, 8, 4, 4], final_scale=80,
    resblock_kernel_sizes=[3, 7, 11], resblock_dilations=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], use_additional_convs=True, bias=True,
    nonlinear_activation="LeakyReLU", nonlinear_activation_params={"negative_slope": 0.1}, use_weight_norm=True, extra_art=False,
    use_ar=True, ar_input=512, ar_hidden=256, ar_output=128)
# (generator overrides, batch_size, batch_max_steps, mel fs, mel hop, LR milestones)
_M = [40000, 80000, 120000, 160000]
_RECIPES = {
    "car": ({}, 64, 2000, 16000, 256, _M),                   # e2w_hifigan_car.yaml:34-58,100-111,134-135
    "e2w": ({}, 32, 8000, 16000, 80, [2 * m for m in _M]),   # e2w_hifigan.yaml
    "mri": ({"in_channels": 358, "upsample_scales": [8, 5, 3, 2], "upsample_kernel_sizes": [16, 10, 6, 4], "final_scale": 240},
            16, 30000, 20000, 256, _M),                      # mri2w_hifigan_car.yaml:34-58,134-135
}
STFT_DEFAULTS = {"fft_sizes": [1024, 2048, 512], "hop_sizes": [120, 240, 50], "win_lengths": [600, 1200, 240], "window": "hann_window"}  # stft_loss.py:131-137


def recipe_train_config(recipe="car", aux="mel", batch=None, fused_optimizers=True):
    """aux: "mel" (what the YAMLs ship) or "stft" (BASELINE config 5's multi-resolution STFT loss with the reference's default resolutions)."""
    g_over, r_batch, r_steps, fs, mel_hop, milestones = _RECIPES[recipe]
    adam = {"lr": 1.0e-4, "betas": [0.5, 0.9], "weight_decay": 0.0}
    sched = {"gamma": 0.5, "milestones": list(milestones)}
    return copy.deepcopy(dict(
        generator_type="HiFiGANGenerator", generator_params=dict(_GENERATOR, **g_over),
        discriminator_type="HiFiGANMultiScaleMultiPeriodDiscriminator",
        discriminator_params=dict(disc_params(), follow_official_norm=True,
                                  scale_discriminator_params=dict(disc_params()["scale_discriminator_params"], downsample_scales=[4, 4, 4, 4, 1])),
        use_stft_loss=aux == "stft", use_mel_loss=aux == "mel", stft_loss_params=dict(STFT_DEFAULTS),
        mel_loss_params=dict(fs=fs, fft_size=1024, hop_size=mel_hop, win_length=None, window="hann", num_mels=80, fmin=0, fmax=11025, log_base=None),
        generator_adv_loss_params={"average_by_discriminators": False}, discriminator_adv_loss_params={"average_by_discriminators": False},
        use_feat_match_loss=True, feat_match_loss_params={"average_by_discriminators": False, "average_by_layers": False, "include_final_outputs": False},
        lambda_aux=45.0, lambda_adv=1.0, lambda_feat_match=2.0, batch_size=batch or r_batch, batch_max_steps=r_steps,
        generator_optimizer_type="Adam", generator_optimizer_params=adam, generator_scheduler_type="MultiStepLR", generator_scheduler_params=sched,
        generator_grad_norm=-1, discriminator_optimizer_type="Adam", discriminator_optimizer_params=adam, discriminator_scheduler_type="MultiStepLR",
        discriminator_scheduler_params=sched, discriminator_grad_norm=-1, generator_train_start_steps=1, discriminator_train_start_steps=0,
        distributed=False, fused_optimizers=fused_optimizers))
