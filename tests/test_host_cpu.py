"""CPU: host-side logic of the product (no kernels run): C-ABI library loads and exports every symbol the header
declares, drop-in modules expose the reference's state-dict keys/shapes, factories parse the reference's config
schema, and the product path refuses CPU tensors instead of silently falling back."""
import ctypes
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from kalle_audio_amd import _lib
    protos = _lib.parse_header()
    assert len(protos) >= 28
    if not os.path.exists(_lib.LIB_PATH):
        from kalle_audio_amd import build
        build.build(verbose=False)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(lib, name), name
    lib.kalle_abi_version.restype = ctypes.c_int
    assert lib.kalle_abi_version() == 2
    lib.kalle_target_arch.restype = ctypes.c_char_p
    assert lib.kalle_target_arch() == b"gfx950"
    # the ctypes mirrors match the C structs of the header: sizes and every field offset, as the C compiler lays them out
    import subprocess
    import tempfile
    pairs = (("kalle_gemm_epilogue", _lib.GemmEpilogue), ("kalle_act", _lib.Act), ("kalle_conv_epilogue", _lib.ConvEpilogue),
             ("kalle_llama_layer", _lib.LlamaLayer), ("kalle_wgrad_problem", _lib.WgradProblem))
    lines = []
    for cname, cls in pairs:
        lines.append(f'printf("{cname} %zu", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf(" %zu", offsetof({cname}, {fname}));')
        lines.append('printf("\\n");')
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "kalle_hip.h"\nint main(void) {\n' + "\n".join(lines) + "\nreturn 0; }\n"
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")], check=True)
        out = subprocess.run([os.path.join(d, "s")], capture_output=True, text=True, check=True).stdout.split("\n")
    for (cname, cls), line in zip(pairs, out):
        got = line.split()
        want = [cname, str(ctypes.sizeof(cls))] + [str(getattr(cls, f).offset) for f, _ in cls._fields_]
        assert got == want, (got, want)


def test_missing_library_fails_loudly(monkeypatch):
    from kalle_audio_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libkalle_hip.so")
    with pytest.raises(_lib.KalleError):
        _lib.load()


def _inventory():
    with open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("gtype", ["prepend", "adaLN"])
def test_dit_state_dict_matches_reference(gtype):
    from kalle_audio_amd.stable_audio_tools.models.dit import DiffusionTransformer
    m = DiffusionTransformer(io_channels=16, embed_dim=128, depth=2, num_heads=2, cond_token_dim=64,
                             project_cond_tokens=True, global_cond_dim=32, prepend_cond_dim=24,
                             transformer_type="continuous_transformer", global_cond_type=gtype)
    mine = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert mine == _inventory()[f"dit_{gtype}"]


def _ae_cfg():
    return {"model_type": "autoencoder", "sample_rate": 16000, "sample_size": 4096, "audio_channels": 2,
            "model": {"encoder": {"type": "oobleck", "config": {"in_channels": 2, "channels": 8, "c_mults": [1, 2, 4],
                                                               "strides": [2, 4, 5], "latent_dim": 8, "use_snake": True}},
                      "decoder": {"type": "oobleck", "config": {"out_channels": 2, "channels": 8, "c_mults": [1, 2, 4],
                                                               "strides": [2, 4, 5], "latent_dim": 4, "use_snake": True,
                                                               "final_tanh": True}},
                      "bottleneck": {"type": "vae"}, "latent_dim": 4, "downsampling_ratio": 40, "io_channels": 2}}


def test_autoencoder_state_dict_and_factory():
    import kalle_audio_amd
    kalle_audio_amd.install()
    from stable_audio_tools.models.factory import create_model_from_config  # resolves to the drop-in
    ae = create_model_from_config(_ae_cfg())
    mine = {k: list(v.shape) for k, v in ae.state_dict().items()}
    assert mine == _inventory()["oobleck_autoencoder"]
    assert ae.downsampling_ratio == 40 and ae.latent_dim == 4


def test_diffusion_cond_factory_and_wrapper_routing():
    from kalle_audio_amd.stable_audio_tools.models.factory import create_model_from_config
    cfg = {"model_type": "diffusion_cond", "sample_rate": 16000, "sample_size": 4096,
           "model": {"io_channels": 16,
                     "conditioning": {"configs": [], "cond_dim": 64},
                     "diffusion": {"type": "dit", "cross_attention_cond_ids": ["prompt"], "global_cond_ids": ["secs"],
                                   "config": {"io_channels": 16, "embed_dim": 128, "depth": 1, "num_heads": 2,
                                              "cond_token_dim": 64, "global_cond_dim": 32,
                                              "transformer_type": "continuous_transformer"}}}}
    w = create_model_from_config(cfg)
    assert w.diffusion_objective == "v" and w.min_input_length == 1
    c = {"prompt": (torch.zeros(2, 5, 64), torch.ones(2, 5, dtype=torch.bool)), "secs": (torch.zeros(2, 32), None)}
    r = w.get_conditioning_inputs(c)
    assert r["cross_attn_cond"].shape == (2, 5, 64) and r["global_cond"].shape == (2, 32)
    # DiTWrapper halves every parameter at construction (models/diffusion.py:505-507)
    g = w.model.model.transformer.layers[0].pre_norm.gamma
    assert torch.allclose(g, torch.full_like(g, 0.5))


def test_cpu_tensors_are_refused_not_emulated():
    from kalle_audio_amd.stable_audio_tools.models.transformer import LayerNorm, TransformerBlock
    with pytest.raises(RuntimeError, match="GPU only"):
        LayerNorm(64)(torch.zeros(2, 3, 64))
    with pytest.raises(RuntimeError, match="GPU only"):
        TransformerBlock(128)(torch.zeros(1, 4, 128))


def test_unsupported_reference_options_raise():
    from kalle_audio_amd.stable_audio_tools.models import transformer as T
    from kalle_audio_amd.stable_audio_tools.models.dit import DiffusionTransformer
    with pytest.raises(NotImplementedError):
        T.Attention(128, natten_kernel_size=7)
    with pytest.raises(ValueError):
        T.Attention(128, qk_norm="rms")
    with pytest.raises(NotImplementedError):
        T.TransformerBlock(128, remove_norms=True)
    with pytest.raises(NotImplementedError):
        DiffusionTransformer(transformer_type="x-transformers")


def test_optional_variants_keep_the_reference_parameter_names():
    """Attention(qk_norm="ln") and DecoderBlock(use_nearest_upsample=True): the parameter names / shapes of the reference's
    modules (transformer.py:305-307 q_norm / k_norm LayerNorm(64); autoencoders.py:87-96 layers.1.1.weight_{g,v}, no bias)"""
    from kalle_audio_amd.stable_audio_tools.models import transformer as T
    from kalle_audio_amd.stable_audio_tools.models.autoencoders import DecoderBlock
    a = dict(T.Attention(128, dim_context=64, qk_norm="ln").named_parameters())
    assert {k: tuple(v.shape) for k, v in a.items() if "_norm" in k} == {
        "q_norm.weight": (64,), "q_norm.bias": (64,), "k_norm.weight": (64,), "k_norm.bias": (64,)}
    assert not any("_norm" in k for k, _ in T.Attention(128, qk_norm="l2").named_parameters())
    d = {k: tuple(v.shape) for k, v in DecoderBlock(32, 16, stride=4, use_nearest_upsample=True).named_parameters()}
    assert d["layers.1.1.weight_v"] == (16, 32, 8) and d["layers.1.1.weight_g"] == (16, 1, 1) and "layers.1.1.bias" not in d
    # round 3: TransformerBlock(conformer=True) (transformer.py:550-567) and the position embeddings of a ContinuousTransformer
    # (transformer.py:45-87, 733-739) - names / shapes as the reference's own modules register them
    b = {k: tuple(v.shape) for k, v in T.TransformerBlock(128, conformer=True, causal=True).named_parameters() if "conformer" in k}
    assert b == {"conformer.in_norm.gamma": (128,), "conformer.pointwise_conv.weight": (128, 128, 1),
                 "conformer.glu.proj.weight": (256, 128), "conformer.glu.proj.bias": (256,),
                 "conformer.depthwise_conv.weight": (128, 1, 17), "conformer.mid_norm.gamma": (128,),
                 "conformer.pointwise_conv_2.weight": (128, 128, 1)}
    ct = T.ContinuousTransformer(dim=128, depth=1, use_sinusoidal_emb=True)
    assert tuple(ct.pos_emb.scale.shape) == (1,) and "pos_emb.inv_freq" not in ct.state_dict()
    ct = T.ContinuousTransformer(dim=128, depth=1, use_abs_pos_emb=True, abs_pos_emb_max_length=50)
    assert tuple(ct.pos_emb.emb.weight.shape) == (50, 128)


def test_reference_yaml_and_accelerate_configs_parse():
    """configs/*.yaml schema - parsed by the trainer's config reader unchanged."""
    from kalle_audio_amd.config import load_experiment_config, load_accelerate_config
    cfg = load_experiment_config(os.path.join(ROOT, "tests", "golden", "example_experiment.yaml"))
    assert cfg["model"]["latent_dim"] == 512 and cfg["gradient_accumulation_steps"] == 2
    acc = load_accelerate_config(os.path.join(ROOT, "tests", "golden", "example_accelerate.yaml"))
    assert acc["distributed_type"] in ("MULTI_GPU", "MULTI_CPU") and acc["num_processes"] == 1


def test_inverse_lr_schedule():
    from kalle_audio_amd.stable_audio_tools.training.utils import InverseLR
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1.0)
    s = InverseLR(opt, inv_gamma=10.0, power=0.5, warmup=0.9)
    lrs = []
    for _ in range(3):
        opt.step()
        s.step()
        lrs.append(opt.param_groups[0]["lr"])
    exp = [(1 - 0.9 ** (e + 1)) * (1 + e / 10.0) ** -0.5 for e in (1, 2, 3)]
    assert all(abs(a - b) < 1e-6 for a, b in zip(lrs, exp))


@pytest.mark.parametrize("tag", ["amp1_causal", "amp2_same"])
def test_melvae_state_dict_matches_reference(tag):
    """backup/flows.py BigVGANFlowVAE: same keys / shapes as the reference module built from the same hyper-parameters"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import golden_util as gu
    from kalle_audio_amd.flows import BigVGANFlowVAE
    m = BigVGANFlowVAE(gu.MELVAE_CONFIGS[tag])
    mine = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert mine == _inventory()[f"melvae_{tag}"]
    with pytest.raises(RuntimeError):
        m.extract_latents(torch.zeros(1, 1, 64))      # CPU tensors are refused, not emulated
    m.remove_weight_norm()
    assert "conv_pre.weight" in m.state_dict() and "conv_pre.weight_g" not in m.state_dict()


def _tiny_llama_dir(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import golden_util as gu
    d = tmp_path / "llama"
    d.mkdir()
    (d / "config.json").write_text(json.dumps(dict(gu.LLASA_CONFIG["llama"], model_type="llama")))
    return str(d), gu


class _Tok:
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n


def test_llasa_state_dict_matches_reference(tmp_path):
    """model_sigmaVAE.Llasa drop-in: HF key names / shapes incl. the tied lm_head, fused q/k/v and up/gate parameters split
    and merged by the state-dict hooks, resize_token_embeddings, CPU tensors refused"""
    path, gu = _tiny_llama_dir(tmp_path)
    from kalle_audio_amd.model_sigmaVAE import Llasa
    lc = gu.LLASA_CONFIG
    m = Llasa({"llm_model_name_or_path": path, "latent_dim": lc["latent_dim"], "audio_proj_dim": 128},
              _Tok(lc["tokenizer_len"]), use_flash_attention=False)
    sd = m.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == _inventory()["llasa"]
    assert sd["base_model.lm_head.weight"].data_ptr() == sd["base_model.model.embed_tokens.weight"].data_ptr()
    new = {k: torch.randn_like(v) for k, v in sd.items()}
    new["base_model.lm_head.weight"] = new["base_model.model.embed_tokens.weight"]
    m.load_state_dict(new)
    att = m.base_model.model.layers[1].self_attn
    want = torch.cat([new[f"base_model.model.layers.1.self_attn.{n}_proj.weight"] for n in "qkv"], 0)
    assert torch.equal(att.qkv_proj.weight.data, want)
    mlp = m.base_model.model.layers[0].mlp
    want = torch.cat([new["base_model.model.layers.0.mlp.up_proj.weight"],
                      new["base_model.model.layers.0.mlp.gate_proj.weight"]], 0)
    assert torch.equal(mlp.up_gate_proj.weight.data, want)
    back = m.state_dict()
    assert all(torch.equal(back[k], new[k]) for k in new)
    assert m.vocab_size == lc["tokenizer_len"] and m.hidden_size == 128
    b = {k: torch.from_numpy(v) for k, v in gu.llasa_batch(lc, 40).items()}
    with pytest.raises(RuntimeError):
        m(b["input_ids"], b["audio_latents"], b["audio_distribution_l"], b["ids_mask"], b["audio_mask"],
          b["target_mask"], b["end_mask"])


def test_ema_schedule():
    """decay schedule of the reference's EMA configuration (training/diffusion.py:240-248)"""
    from kalle_audio_amd.engine import EMASchedule
    s = EMASchedule(beta=0.9999, power=3 / 4, update_every=1, update_after_step=1)
    acts = [s.next() for _ in range(6)]
    assert acts[0] == "copy" and acts[1] == "copy"           # steps 0, 1 <= update_after_step: plain copies
    assert acts[2] == "copy"                                  # first step past the warm-up initialises the average
    assert abs(acts[3] - (1 - (1 + 2) ** -0.75)) < 1e-12      # epoch = step_count - update_after_step - 1 = 2
    assert abs(acts[4] - (1 - (1 + 3) ** -0.75)) < 1e-12
    big = EMASchedule(beta=0.9999, power=3 / 4, update_every=1, update_after_step=1)
    big.step, big.initted = 10 ** 7, True
    assert big.next() == 0.9999
    sk = EMASchedule(update_every=10, update_after_step=0)
    assert [sk.next() is None for _ in range(11)] == [False] + [True] * 9 + [False]


def test_shim_pythonpath_needs_no_edited_line(tmp_path):
    """`PYTHONPATH=<repo>/shim python script.py` - the script's own directory holds modules called `model`,
    `model_sigmaVAE`, `flows` and a package `stable_audio_tools` (as the reference's root does) that must NOT be the ones
    imported; the script itself has no kalle-specific line"""
    import subprocess
    root = ROOT
    for name in ("model.py", "model_sigmaVAE.py", "flows.py"):
        (tmp_path / name).write_text("raise RuntimeError('the reference module was imported')\n")
    (tmp_path / "stable_audio_tools").mkdir()
    (tmp_path / "stable_audio_tools" / "__init__.py").write_text("raise RuntimeError('the reference package was imported')\n")
    (tmp_path / "train.py").write_text(
        "from model import Llasa as A\n"
        "from model_sigmaVAE import Llasa as B\n"
        "from stable_audio_tools import create_model_from_config\n"
        "from stable_audio_tools.models.utils import load_ckpt_state_dict\n"
        "from stable_audio_tools.inference.generation import generate_diffusion_cond\n"
        "from flows import BigVGANFlowVAE\n"
        "print(A.__module__, B.__module__, create_model_from_config.__module__, BigVGANFlowVAE.__module__)\n")
    env = dict(os.environ, PYTHONPATH=os.path.join(root, "shim"))
    r = subprocess.run([sys.executable, "train.py"], cwd=tmp_path, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split() == ["kalle_audio_amd.model", "kalle_audio_amd.model_sigmaVAE",
                                "kalle_audio_amd.stable_audio_tools.models.factory", "kalle_audio_amd.flows"], r.stdout
    r = subprocess.run([sys.executable, "train.py"], cwd=tmp_path, capture_output=True, text=True, timeout=300,
                       env=dict(env, KALLE_SHIM="0"))
    assert r.returncode != 0 and "the reference module was imported" in r.stderr      # switched off: the directory wins again


def test_gemm_workspace_is_capped_and_reused(monkeypatch):
    """ops._workspace: a request beyond the 1 GiB cap must be served by ONE capped buffer that later calls reuse (round 3: it was
    re-allocated - and the old one kept - on every call, 11 GB per train step at 32 clips per GPU)"""
    from kalle_audio_amd import ops
    made = []

    class _Stream:
        cuda_stream = 7

    monkeypatch.setattr(ops.torch.cuda, "current_stream", lambda device=None: _Stream())
    monkeypatch.setattr(ops.torch, "empty", lambda n, device=None, dtype=None: _Fake(n, made))
    monkeypatch.setattr(ops, "_WS", {})
    monkeypatch.setattr(ops, "_WS_KEEP", [])
    dev = torch.device("cuda", 0)
    a = ops._workspace(dev, 4 * 4032 * 12288 * 8)          # 1.58 GB asked
    b = ops._workspace(dev, 4 * 4032 * 12288 * 8)
    c = ops._workspace(dev, 1 << 20)
    assert a is b is c and made == [1 << 30] and ops._WS_KEEP == []
    d = ops._workspace(torch.device("cuda", 1), 1 << 20)
    assert d is not a and made == [1 << 30, 64 << 20]


class _Fake:
    def __init__(self, n, log):
        self.n = n
        log.append(n)

    def numel(self):
        return self.n


def test_layernorm_backward_workgroup_count():
    """kalle_layernorm_bwd_parts is host arithmetic (no GPU call): the number of partial rows the caller must provide = the launch's
    workgroup count - 8 rows per workgroup below the 1024-workgroup cap (B = 16 per GPU: 2016 rows), the cap above it"""
    from kalle_audio_amd import _lib
    lib = _lib.load()
    if os.environ.get("KALLE_LN_BWD_RPB"):
        pytest.skip("experiment switch set")
    assert lib.kalle_layernorm_bwd_parts(1) == 1
    assert lib.kalle_layernorm_bwd_parts(8) == 1 and lib.kalle_layernorm_bwd_parts(9) == 2
    assert lib.kalle_layernorm_bwd_parts(2016) == 252
    assert lib.kalle_layernorm_bwd_parts(8064) == 1008
    assert lib.kalle_layernorm_bwd_parts(32256) == 1024 and lib.kalle_layernorm_bwd_parts(10 ** 6) == 1024
