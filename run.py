"""Entry script with the flow of the reference's run.py (reference run.py:19-104): YAML ->
data module -> versioned logger -> seed -> model (optionally resumed from the previous version's
last.safetensors) -> checkpoint callback -> fit -> rank-0 test.

    python run.py --config configs/butterfly_ddpm.yaml [--synthetic] [--max-steps N] [--save-dir DIR]
    python -m torch.distributed.run --nproc-per-node 8 run.py --config ...   (data parallel)

`--synthetic` replaces the ImageFolder on disk by ImageFolder-shaped random batches; `--save-dir`
overrides `log_config.save_dir`.
"""
import argparse
import datetime
import os
import shutil

import torch
import torch.distributed as dist
import yaml

from vision_amd import chose_model, get_model
from vision_amd.trainer import DirLogger, ImageFolderDataModule, SyntheticImageFolder, Trainer, seed_everything
from vision_amd.utils import SafetensorsCheckpoint


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", dest="filename", default="configs/flowers_ddpm.yaml")
    parser.add_argument("--synthetic", action="store_true")
    parser.add_argument("--max-steps", type=int, default=-1)
    parser.add_argument("--save-dir", default=None)
    args = parser.parse_args(argv)
    with open(args.filename, "r") as f:
        config = yaml.safe_load(f)
    if args.save_dir is not None:
        config["log_config"]["save_dir"] = args.save_dir

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)) % max(1, torch.cuda.device_count()))
        # a collective some rank never joins ends the job with an error after this long instead of hanging
        dist.init_process_group(os.environ.get("VAMD_DIST_BACKEND", "nccl"),
                                timeout=datetime.timedelta(seconds=int(os.environ.get("VAMD_DIST_TIMEOUT_S", "600"))))
    rank = dist.get_rank() if dist.is_initialized() else 0

    data_cls = SyntheticImageFolder if args.synthetic else ImageFolderDataModule
    data_module = data_cls(**config["data_config"])
    data_module.setup()                                   # before seeding, as in the reference (run.py:32 vs :44)
    if rank == 0:
        print(f"Number of classes: {data_module.num_classes}")

    logger = DirLogger(save_dir=config["log_config"]["save_dir"], name=config["log_config"]["name"])
    seed_everything(config["model_config"]["manual_seed"], True)
    model = get_model(config["model_config"]["name"], config["model_config"])

    if config["log_config"]["checkpoint"]:
        try:
            prev = os.path.join(os.path.dirname(logger.log_dir), f"version_{logger.version - 1}")
            last = os.path.join(prev, "checkpoints", "last.safetensors")
            if os.path.exists(last):
                model = model.__class__.load_from_checkpoint(last, config=config["model_config"])
                print(f"Loaded checkpoint from {last}")
            else:
                print(f"No checkpoint found at {last}")
        except Exception as e:  # the reference swallows resume errors too (run.py:58)
            print(f"Error loading checkpoint: {e}")

    # latent diffusion: `first_stage_config` names the YAML of the autoencoder whose latents are diffused and
    # `first_stage_checkpoint` (optional) its trained weights; without a checkpoint the first stage keeps its seeded
    # initialisation (smoke runs).  The reference has no entry for models/ldm.py at all.
    fs_yaml = config["model_config"].get("first_stage_config")
    if fs_yaml is not None:
        with open(fs_yaml, "r") as f:
            fs_cfg = yaml.safe_load(f)["model_config"]
        fs_ckpt = config["model_config"].get("first_stage_checkpoint")
        fs_cls = chose_model[fs_cfg["name"]]
        first = fs_cls.load_from_checkpoint(fs_ckpt, config=fs_cfg) if fs_ckpt else get_model(fs_cfg["name"], fs_cfg)
        model.set_first_stage(first)

    checkpoint = SafetensorsCheckpoint(config_path=args.filename, dirpath=os.path.join(logger.log_dir, "checkpoints"),
                                       monitor="VL", mode="min", save_top_k=2, save_last=True,
                                       filename="epoch={epoch}-VL={VL:.2f}")
    trainer = Trainer(enable_checkpointing=False, logger=logger, callbacks=[checkpoint], log_every_n_steps=1,
                      enable_progress_bar=True, max_steps=args.max_steps, **config["trainer_config"])
    trainer.fit(model, datamodule=data_module)

    if rank == 0:
        # rank 0 alone, like the reference (run.py:93-96): a trainer that takes no part in the process group
        Trainer(devices=1, num_nodes=1, logger=logger, distributed=False,
                precision=config["trainer_config"].get("precision", "16-mixed")) \
            .test(model, data_module.test_dataloader(), verbose=True)
        if os.path.exists("lightning_logs"):
            shutil.rmtree("lightning_logs")
    if dist.is_initialized():
        dist.barrier()
        if world > 1:
            dist.destroy_process_group()
    return model, logger


if __name__ == "__main__":
    main()
