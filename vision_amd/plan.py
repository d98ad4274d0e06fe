"""Static layer plan of the DDPM UNet, derived from the reference's YAML `model_config`.

The reference builds its module lists imperatively in `DDPMModel.__init__`
(reference models/ddpm.py:160-204); here the same decisions produce a flat list of `LayerSpec`
records that both the parameter containers (ddpm.py) and the HIP execution engine (engine.py)
are generated from, so that they cannot disagree.
"""
from dataclasses import dataclass
from typing import List

REQUIRED_KEYS = ("h_dims", "patch_size", "attn_res", "down_index", "with_conv", "dropout",
                 "latent_dim", "sampling_period", "diffusion_steps", "grid")


@dataclass(frozen=True)
class LayerSpec:
    section: str      # 'Downsampling' | 'Middle' | 'Upsampling'
    index: int        # position inside the section's ModuleList
    kind: str         # 'res' | 'attn' | 'down' | 'up'
    cin: int
    cout: int
    res_in: int       # spatial extent of the module's input
    res_out: int
    dropout: float = 0.0

    @property
    def prefix(self):
        return f"{self.section}.{self.index}"


@dataclass(frozen=True)
class UNetPlan:
    layers: List[LayerSpec]
    ch: int               # width of the sinusoidal embedding (h_dims[1])
    time_features: int    # 4 * ch
    end_channels: int
    patch_size: int
    with_conv: bool

    def section(self, name):
        return [l for l in self.layers if l.section == name]


def _unet_plan(dims, down_flags, attn_at, res, with_conv, p, patch_size) -> UNetPlan:
    """Module sequence of the time-conditioned UNet (reference models/ddpm.py:160-204; models/ldm.py:170-214 builds the
    same sequence)."""
    levels = len(dims) - 1
    layers: List[LayerSpec] = []

    # The reference tracks the feature resolution in a variable that becomes a float after the
    # first halving and tests `f_res in attn_res`; int/float equality makes 112.0 match 112.
    pos = 0
    width = dims[0]
    for lvl in range(levels):
        width_in, width = dims[lvl], dims[lvl + 1]
        layers.append(LayerSpec("Downsampling", pos, "res", width_in, width, int(res), int(res)))
        pos += 1
        if res in attn_at:                                   # checked before the level's DownBlock
            layers.append(LayerSpec("Downsampling", pos, "attn", width, width, int(res), int(res)))
            pos += 1
        if down_flags[lvl + 1] == 1:
            # conv 3x3 stride 2 pad 1 (== res/2 for even res); AvgPool2d((2, 2)) floors
            half = (int(res) - 1) // 2 + 1 if with_conv else int(res) // 2
            layers.append(LayerSpec("Downsampling", pos, "down", width, width, int(res), half))
            pos += 1
            res = res / 2

    r = int(res)
    layers.append(LayerSpec("Middle", 0, "res", width, width, r, r, p))
    layers.append(LayerSpec("Middle", 1, "attn", width, width, r, r))
    layers.append(LayerSpec("Middle", 2, "res", width, width, r, r, p))

    rdims, rflags = dims[::-1], down_flags[::-1]
    pos = 0
    for lvl in range(levels):
        width_in, width = rdims[lvl], rdims[lvl + 1]
        if rflags[lvl] == 1:
            layers.append(LayerSpec("Upsampling", pos, "up", width_in, width_in, int(res), int(res) * 2))
            pos += 1
            res = res * 2
        layers.append(LayerSpec("Upsampling", pos, "res", width_in, width, int(res), int(res)))
        pos += 1
        if res in attn_at:                                   # checked after the level's UpBlock
            layers.append(LayerSpec("Upsampling", pos, "attn", width, width, int(res), int(res)))
            pos += 1
    return UNetPlan(layers=layers, ch=dims[1], time_features=4 * dims[1], end_channels=width,
                    patch_size=int(patch_size), with_conv=with_conv)


def make_plan(config) -> UNetPlan:
    for k in REQUIRED_KEYS:
        if k not in config:
            raise KeyError(k)          # the reference indexes config[...] directly (models/ddpm.py:142-153)
    return _unet_plan(list(config["h_dims"]), list(config["down_index"]), list(config["attn_res"]), config["patch_size"],
                      bool(config["with_conv"]), float(config["dropout"]), config["patch_size"])


LDM_REQUIRED_KEYS = ("h_dims", "in_res", "attn_res", "down_index", "with_conv", "dropout", "diffusion_steps",
                     "sampling_period", "grid")


def make_ldm_plan(config) -> UNetPlan:
    """The latent-space UNet (reference models/ldm.py:139-221): the DDPM module sequence over `in_res` x `in_res` maps
    of h_dims[0] channels.  The reference reads these keys with config.get and fails later on a None; a missing key is
    reported here."""
    for k in LDM_REQUIRED_KEYS:
        if config.get(k) is None:
            raise KeyError(k)
    return _unet_plan(list(config["h_dims"]), list(config["down_index"]), list(config["attn_res"]), config["in_res"],
                      bool(config["with_conv"]), float(config["dropout"]), config["in_res"])


VQ_REQUIRED_KEYS = ("h_dims", "patch_size", "num_embed", "embed_dim", "sampling_period", "attn_res", "down_index")


def make_vq_plan(config) -> UNetPlan:
    """Layer plan of the VQ-VAE-UNet first stage (reference models/vqvaeunet.py:150-203): per level a ResNetBlock, an
    AttnBlock where the resolution is listed in attn_res and, where down_index says so, a DownBlock; mirrored by UpBlock
    + ResNetBlock (+ AttnBlock) per level; the codebook lookup sits between
    the two halves, the channel-wise Linear + Sigmoid head after them.  The blocks take no time embedding
    (time_features = 0) and always resample with convolutions (models/vqvaeunet.py:161)."""
    for k in VQ_REQUIRED_KEYS:
        if k not in config:
            raise KeyError(k)          # the reference indexes config[...] directly (models/vqvaeunet.py:153-159)
    dims = list(config["h_dims"])
    down_flags = list(config["down_index"])
    attn_at = list(config["attn_res"])
    levels = len(dims) - 1
    layers: List[LayerSpec] = []

    res = config["patch_size"]
    pos = 0
    width = dims[0]
    for lvl in range(levels):
        width_in, width = dims[lvl], dims[lvl + 1]
        layers.append(LayerSpec("Downsampling", pos, "res", width_in, width, int(res), int(res)))
        pos += 1
        if res in attn_at:                                       # checked before the level's DownBlock; `f_res in
            layers.append(LayerSpec("Downsampling", pos, "attn", width, width, int(res), int(res)))   # attn_res` on a float
            pos += 1
        if down_flags[lvl + 1] == 1:
            layers.append(LayerSpec("Downsampling", pos, "down", width, width, int(res), (int(res) - 1) // 2 + 1))
            pos += 1
            res = res / 2
    if width != int(config["embed_dim"]):
        raise ValueError(f"embed_dim {config['embed_dim']} must equal the encoder's last width {width}: the codebook "
                         "rows replace the encoder's output vectors (models/vqvaeunet.py:118,210)")
    rdims, rflags = dims[::-1], down_flags[::-1]
    pos = 0
    for lvl in range(levels):
        width_in, width = rdims[lvl], rdims[lvl + 1]
        if rflags[lvl] == 1:
            layers.append(LayerSpec("Upsampling", pos, "up", width_in, width_in, int(res), int(res) * 2))
            pos += 1
            res = res * 2
        layers.append(LayerSpec("Upsampling", pos, "res", width_in, width, int(res), int(res)))
        pos += 1
        if res in attn_at:                                       # checked after the level's UpBlock
            layers.append(LayerSpec("Upsampling", pos, "attn", width, width, int(res), int(res)))
            pos += 1
    return UNetPlan(layers=layers, ch=0, time_features=0, end_channels=width,
                    patch_size=int(config["patch_size"]), with_conv=True)
