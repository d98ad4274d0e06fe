"""A small stand-in for what the reference's `run.py` gets from pytorch-lightning: the fit /
validate / test loops, the versioned log directory, callbacks and one-process-per-GPU data
parallelism.  Only what `models/ddpm.py` and `run.py` of the reference rely on is implemented.

Loop semantics (Lightning 2.0 defaults, which the reference inherits): training batches in train
mode; after each epoch the LR scheduler steps and `on_train_epoch_end` runs; validation runs in
eval mode under no_grad, its epoch mean of 'VL' feeds `callback_metrics`, then
`on_validation_end` hooks (sampling, checkpointing) fire.
"""
import json
import os
import random
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

from .parallel import DataParallel, world_info

_PRECISIONS = {"16-mixed": "bf16", "bf16-mixed": "bf16", "bf16": "bf16", "32": "fp32", "32-true": "fp32", 32: "fp32"}


def seed_everything(seed: int, workers: bool = False):
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    return seed


class DirLogger:
    """`save_dir/name/version_N` directory layout of Lightning's TensorBoardLogger; scalars are
    appended to metrics.jsonl (tensorboard is optional and not required)."""

    def __init__(self, save_dir, name):
        self.save_dir, self.name = save_dir, name
        root = os.path.join(save_dir, name)
        os.makedirs(root, exist_ok=True)
        taken = [int(d.split("_")[1]) for d in os.listdir(root) if d.startswith("version_") and d.split("_")[1].isdigit()]
        self.version = max(taken) + 1 if taken else 0
        self.log_dir = os.path.join(root, f"version_{self.version}")

    def log_metrics(self, metrics, step):
        os.makedirs(self.log_dir, exist_ok=True)
        with open(os.path.join(self.log_dir, "metrics.jsonl"), "a") as f:
            f.write(json.dumps({"step": step, **{k: float(v) for k, v in metrics.items()}}) + "\n")


class SyntheticImageFolder:
    """ImageFolder-shaped batches without a dataset on disk: x float32 [B,3,P,P] uniform in [0,1)
    (what Resize + ToTensor yield, reference dataloader.py:32-36), y int64 zeros; an 80/10/10
    train/valid/test split like the reference's manual split (dataloader.py:56-64)."""

    def __init__(self, patch_size=224, batch_size=32, length=256, seed=0, device="cpu", **_):
        self.patch_size, self.batch_size, self.length, self.seed, self.device = patch_size, batch_size, length, seed, device
        self.num_classes = 1

    def setup(self, stage=None):
        n_test = int(self.length * 0.1)
        n_val = int(self.length * 0.1)
        self.sizes = {"train": self.length - n_test - n_val, "valid": n_val, "test": n_test}

    def _loader(self, split, rank=0, world=1):
        n = self.sizes[split]
        per_rank = n // world if split == "train" else n
        g = torch.Generator(device="cpu").manual_seed(self.seed + 1000 * ("train", "valid", "test").index(split) + rank)
        for i in range(0, per_rank - self.batch_size + 1, self.batch_size):
            x = torch.rand((self.batch_size, 3, self.patch_size, self.patch_size), generator=g)
            yield x.to(self.device, non_blocking=True), torch.zeros(self.batch_size, dtype=torch.int64, device=self.device)

    def train_dataloader(self, rank=0, world=1):
        return self._loader("train", rank, world)

    def val_dataloader(self):
        return self._loader("valid")

    def test_dataloader(self):
        return self._loader("test")


class ImageFolderDataModule:
    """PIL implementation of the reference's ImageFolderDataModule (dataloader.py:10-82) for the
    use_manual_split layouts: class sub-directories of images, Resize((p, p)) + ToTensor, random
    80/10/10 split, or train/valid/test sub-directories."""

    EXT = (".jpg", ".jpeg", ".png", ".bmp", ".webp")

    def __init__(self, data_dir, json_dir=None, patch_size=224, batch_size=32, num_workers=4, val_split=0.1,
                 test_split=0.1, use_manual_split=False, device="cpu", gpu_transform=None, **_):
        """gpu_transform (default: on when `device` is a HIP device): images are decoded by `num_workers` host
        threads and resized / converted on the GPU, bit-identical to the PIL path (vision_amd.input_pipeline);
        off: every sample is resized and converted on the host, as the reference's DataLoader workers do."""
        self.data_dir, self.patch_size, self.batch_size = data_dir, patch_size, batch_size
        self.val_split, self.test_split, self.use_manual_split, self.device = val_split, test_split, use_manual_split, device
        self.num_workers = num_workers
        self.gpu_transform = gpu_transform            # None: decided by the device the loaders run for
        self._pipeline = None
        self.epoch = 0                 # set by the trainer before each epoch's train_dataloader()

    def _scan(self, root):
        classes = sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d)))
        items = []
        for ci, c in enumerate(classes):
            for dp, _, files in sorted(os.walk(os.path.join(root, c))):
                items += [(os.path.join(dp, f), ci) for f in sorted(files) if f.lower().endswith(self.EXT)]
        return classes, items

    def setup(self, stage=None):
        if self.use_manual_split:
            self.classes, items = self._scan(self.data_dir)
            n = len(items)
            n_test, n_val = int(n * self.test_split), int(n * self.val_split)
            perm = torch.randperm(n).tolist()
            pick = lambda idx: [items[i] for i in idx]
            self.splits = {"train": pick(perm[:n - n_test - n_val]), "valid": pick(perm[n - n_test - n_val:n - n_test]),
                           "test": pick(perm[n - n_test:])}
        else:
            self.splits = {}
            for s in ("train", "valid", "test"):
                self.classes, self.splits[s] = self._scan(os.path.join(self.data_dir, s))
        self.num_classes = len(self.classes)

    def _load(self, path):
        from PIL import Image
        img = Image.open(path).convert("RGB").resize((self.patch_size, self.patch_size), Image.BILINEAR)
        return torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div_(255.0)

    @staticmethod
    def shard(order, rank, world):
        """Rank `rank`'s share of `order`, the same length on every rank: the list is padded with its own
        head up to a multiple of `world` and dealt round-robin (what torch's DistributedSampler does with
        drop_last=False, which Lightning injects for the reference under DDP).  Ranks that ran a different
        number of batches would leave the gradient all-reduce of the longest one without partners."""
        if world <= 1 or not order:
            return list(order)
        total = -(-len(order) // world) * world
        padded = list(order)
        while len(padded) < total:
            padded += order[:total - len(padded)]
        return padded[rank:total:world]

    def _loader(self, split, shuffle, rank=0, world=1):
        items = self.splits[split]
        if shuffle:
            # every rank must deal from the same permutation: draw it from a generator seeded alike
            # (the global seed `seed_everything` set - model_config.manual_seed in run.py, the same on every rank - and
            # the trainer's epoch: another seed gives another order, a resumed run does not replay epoch 0's)
            g = torch.Generator().manual_seed((torch.initial_seed() * 1000003 + self.epoch) % (2 ** 63))
            order = torch.randperm(len(items), generator=g).tolist()
        else:
            order = list(range(len(items)))
        order = self.shard(order, rank, world)
        chunks = [[items[j] for j in order[i:i + self.batch_size]] for i in range(0, len(order), self.batch_size)]
        on_gpu = torch.device(self.device).type == "cuda"
        if self.gpu_transform and not on_gpu:
            raise ValueError("gpu_transform needs the data module's device to be a HIP device (cuda:N)")
        if on_gpu if self.gpu_transform is None else self.gpu_transform:
            if self._pipeline is None or self._pipeline.device != torch.device(self.device):
                from .input_pipeline import GpuImagePipeline
                self._pipeline = GpuImagePipeline(self.patch_size, self.device, workers=max(1, int(self.num_workers)))
            xs = self._pipeline.batches([[p for p, _ in chunk] for chunk in chunks])
            for chunk, x in zip(chunks, xs):
                yield x, torch.tensor([c for _, c in chunk], dtype=torch.int64).to(self.device, non_blocking=True)
            return
        for chunk in chunks:
            x = torch.stack([self._load(p) for p, _ in chunk])
            y = torch.tensor([c for _, c in chunk], dtype=torch.int64)
            yield x.to(self.device), y.to(self.device)

    def train_dataloader(self, rank=0, world=1):
        return self._loader("train", True, rank, world)

    def val_dataloader(self):
        return self._loader("valid", False)

    def test_dataloader(self):
        return self._loader("test", False)


def _with_lookahead(loader, announce):
    """The loader's batches, each yielded after `announce(next batch)` has been called (a model that prepares part of a
    step ahead of time, LatentDiffusion.encode_ahead); announce=None: the loader as it is."""
    if announce is None:
        yield from loader
        return
    it = iter(loader)
    try:
        cur = next(it)
    except StopIteration:
        return
    for nxt in it:
        announce(nxt)
        yield cur
        cur = nxt
    yield cur


class Trainer:
    """fit / test loops with the handful of `pl.Trainer` arguments the reference's YAMLs use
    (accelerator, devices, max_epochs, precision) plus max_steps / limit_* for smoke runs."""

    def __init__(self, logger: Optional[DirLogger] = None, callbacks=(), max_epochs=1, precision="16-mixed",
                 accelerator="gpu", devices=1, max_steps=-1, limit_val_batches=None, log_every_n_steps=1,
                 enable_checkpointing=False, enable_progress_bar=False, num_nodes=1, distributed=True, graph_step=False, **_):
        """distributed=False: a single-process trainer even when a process group is up (no broadcasts, no
        scalar all-reduce) - what the reference's rank-0 `Trainer(devices=1).test(...)` after fit is
        (run.py:93-96); with the group's collectives it would wait for ranks that never call them."""
        self.logger, self.callbacks = logger, list(callbacks)
        self.max_epochs, self.max_steps, self.limit_val_batches = max_epochs, max_steps, limit_val_batches
        self.precision = _PRECISIONS.get(precision, "bf16")
        self.current_epoch, self.global_step = 0, 0
        self.callback_metrics = {}
        self.distributed = bool(distributed)
        self.graph_step = bool(graph_step)
        self.global_rank, self.world_size = world_info() if self.distributed else (0, 1)
        self.lightning_module = None
        self._epoch_sums = {}
        self.dp: Optional[DataParallel] = None

    @property
    def is_global_zero(self):
        return self.global_rank == 0

    def record(self, name, value, on_step=False, on_epoch=True, sync_dist=False, **_):
        v = value.detach() if torch.is_tensor(value) else torch.tensor(float(value))
        if sync_dist and self.dp is not None and self.world_size > 1:
            v = self.dp.all_reduce_scalar_mean(v)
        s, c = self._epoch_sums.get(name, (0.0, 0))
        self._epoch_sums[name] = (s + v, c + 1)          # stays on the device: no host sync per step

    def _flush(self, names):
        out = {}
        for n in names:
            if n in self._epoch_sums:
                s, c = self._epoch_sums.pop(n)
                out[n] = float(s) / c
        return out

    def _attach(self, model):
        # ranks beyond the device count wrap around (several gloo ranks rehearsing on one GPU)
        device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)) % torch.cuda.device_count()) \
            if torch.cuda.is_available() else None
        if device is None:
            raise RuntimeError("vision_amd needs an MI355X (HIP device); no CPU path exists")
        torch.cuda.set_device(device)
        model.to(device)
        model.compute_dtype = {"bf16": torch.bfloat16, "fp32": torch.float32}[self.precision]
        model.trainer, model.logger = self, self.logger
        self.lightning_module = model
        if self.distributed:
            self.dp = DataParallel(model)
        else:
            # no exchange of any kind: detach a reducer an earlier distributed trainer may have left
            self.dp = None
            model.grad_reducer = None
            model._ensure_engine().reducer = None
        return device

    def fit(self, model, datamodule):
        device = self._attach(model)
        datamodule.device = device
        datamodule.setup()
        self.datamodule = datamodule                  # models read trainer.datamodule (reference models/vqvaeunet.py:294)
        cfg = model.configure_optimizers()
        if cfg is None:
            # the reference returns None when scheduler_gamma is present but null (models/ddpm.py:347-352),
            # which Lightning reads as "no optimizer"; training cannot proceed, so say why
            raise ValueError("configure_optimizers() returned None (scheduler_gamma is null in model_config): "
                             "set a number, or remove the key to train with the bare optimizer")
        if isinstance(cfg, tuple):
            (opt,), scheds = cfg
        else:
            opt, scheds = cfg, []
        for cb in self.callbacks:
            if hasattr(cb, "on_train_start") and self.is_global_zero:
                cb.on_train_start(self, model)
        # VAMD_GRAPH=1 (or Trainer(graph_step=True)): the training step is captured into a HIP graph after two eager steps
        # and replayed (vision_amd.graphed: one host call per step; ragged last batches still run eagerly)
        stepper, graph_shape = None, None
        if getattr(self, "graph_step", False) or os.environ.get("VAMD_GRAPH") == "1":
            from .graphed import GraphedTrainStep
            stepper = GraphedTrainStep(model, opt)
        done = False
        for epoch in range(self.max_epochs):
            self.current_epoch = epoch
            if hasattr(datamodule, "epoch"):
                datamodule.epoch = epoch
            model.train()
            for i, batch in enumerate(_with_lookahead(datamodule.train_dataloader(self.global_rank, self.world_size),
                                                      getattr(model, "encode_ahead", None))):
                if self.dp is not None:
                    self.dp.broadcast_buffers()
                shape = tuple(batch[0].shape)
                if stepper is not None and (graph_shape is None or shape == graph_shape):
                    graph_shape = shape
                    loss = stepper(batch, i)
                else:
                    loss = model.training_step(batch, i)
                    loss.backward()
                    opt.step()
                self.global_step += 1
                if 0 < self.max_steps <= self.global_step:
                    done = True
                    break
            for s in scheds:
                s.step()
            train_metrics = self._flush(["TL"])
            if hasattr(model, "on_train_epoch_end"):
                model.on_train_epoch_end()
            model.eval()
            with torch.no_grad():
                for i, batch in enumerate(datamodule.val_dataloader()):
                    if self.limit_val_batches is not None and i >= self.limit_val_batches:
                        break
                    model.validation_step(batch, i)
            self.callback_metrics.update(train_metrics)
            self.callback_metrics.update(self._flush(["VL"]))
            if self.logger is not None and self.is_global_zero:
                self.logger.log_metrics({**self.callback_metrics, "lr": opt.param_groups[0]["lr"]}, self.global_step)
            with torch.no_grad():
                for hook in ("on_validation_epoch_end", "on_validation_end"):      # Lightning's order
                    if hasattr(model, hook):
                        getattr(model, hook)()
                if self.is_global_zero:
                    for cb in self.callbacks:
                        if hasattr(cb, "on_validation_end"):
                            cb.on_validation_end(self, model)
            if done:
                break
        return self

    def test(self, model, dataloader, verbose=True):
        if self.lightning_module is not model:
            self._attach(model)
        model.eval()
        with torch.no_grad():
            for i, batch in enumerate(dataloader):
                model.test_step(batch, i)
        out = self._flush(["TeL"])
        if verbose and self.is_global_zero:
            print(out)
        return [out]
