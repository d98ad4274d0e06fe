"""The three DDPM workloads the reference ships as YAML (configs/{flowers,butterfly,ffhq}_ddpm.yaml),
as Python dicts with the same four sections and keys, for benchmarks and tests.  `run.py` reads the
reference's own YAML files unchanged; `write_yaml` materialises one of these in that schema."""
import copy

_COMMON_MODEL = dict(name="ddpm", input_channels=3, learning_rate=1e-4, manual_seed=42, with_conv=True,
                     dropout=0.1, scheduler_gamma=0.99)


def _preset(h_dims, down_index, attn_res, patch, sampling_period, weight_decay, diffusion_steps, grid,
            data_dir, batch_size, devices, max_epochs, log_name, checkpoint):
    return {
        "model_config": dict(_COMMON_MODEL, h_dims=h_dims, down_index=down_index, attn_res=attn_res,
                             latent_dim=[3, patch, patch], sampling_period=sampling_period,
                             weight_decay=weight_decay, diffusion_steps=diffusion_steps, patch_size=patch, grid=grid),
        "data_config": dict(patch_size=patch, data_dir=data_dir, batch_size=batch_size, num_workers=4,
                            use_manual_split=True),
        "trainer_config": dict(accelerator="gpu", devices=devices, max_epochs=max_epochs, precision="16-mixed"),
        "log_config": dict(name=log_name, save_dir="logs/", check_dir="logs/check", checkpoint=checkpoint,
                           model_summary=False),
    }


PRESETS = {
    # 224x224, widths 64..512, attention at 14x14 (Middle only)
    "flowers": _preset([3, 64, 128, 256, 512], [0, 1, 1, 1, 1], [14], 224, 1, 1e-3, 20, 2,
                       "data/flowers/train", 16, 1, 2000, "flower_ddpm", True),
    "butterfly": _preset([3, 64, 128, 256, 512], [0, 1, 1, 1, 1], [14], 224, 20, 1e-5, 50, 2,
                         "data/butterfly", 64, 1, 2000, "butterfly_ddpm", False),
    # 1024x1024, six levels, attention at 32x32 (down and up paths) and 16x16 (Middle)
    "ffhq": _preset([3, 64, 64, 128, 128, 256, 256], [0, 1, 1, 1, 1, 1, 1], [16, 32], 1024, 20, 1e-4, 50, 1,
                    "data/ffhq", 4, 2, 100, "ffhq_ddpm", True),
}


# the first stage the reference ships (configs/butterfly_vqvaeunet.yaml) and a latent-diffusion model over its 256 x 7 x 7
# latents (the reference has no YAML for models/ldm.py; configs/butterfly_ldm.yaml is this dict)
PRESETS["vqvaeunet"] = {
    "model_config": dict(name="vqvaeunet", learning_rate=1e-3, manual_seed=42, h_dims=[3, 64, 64, 128, 128, 256, 256],
                         down_index=[0, 1, 0, 1, 1, 1, 1], patch_size=224, sampling_period=1, embed_dim=256, num_embed=2048,
                         attn_res=[7], weight_decay=1e-3, scheduler_gamma=0.99),
    "data_config": dict(data_dir="data/butterfly", batch_size=16, patch_size=224, num_workers=4, use_manual_split=True),
    "trainer_config": dict(accelerator="gpu", devices=1, max_epochs=500, precision="16-mixed"),
    "log_config": dict(name="butterfly_vqvaeunet", save_dir="logs/", check_dir="logs/check", checkpoint=True,
                       model_summary=False),
}
PRESETS["ldm"] = {
    "model_config": dict(name="ldm", learning_rate=1e-4, manual_seed=42, with_conv=True, dropout=0.1, scheduler_gamma=0.99,
                         h_dims=[256, 256, 256], down_index=[0, 0, 0], attn_res=[7], in_res=7, sampling_period=20,
                         weight_decay=1e-5, diffusion_steps=50, grid=2),
    "data_config": dict(patch_size=224, data_dir="data/butterfly", batch_size=64, num_workers=4, use_manual_split=True),
    "trainer_config": dict(accelerator="gpu", devices=1, max_epochs=2000, precision="16-mixed"),
    "log_config": dict(name="butterfly_ldm", save_dir="logs/", check_dir="logs/check", checkpoint=False, model_summary=False),
}


def get_preset(name):
    return copy.deepcopy(PRESETS[name])


def write_yaml(name, path):
    import yaml
    with open(path, "w") as f:
        yaml.safe_dump(get_preset(name), f, sort_keys=False)
    return path
