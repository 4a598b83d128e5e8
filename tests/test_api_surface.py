"""The reference's user-facing surface (SURVEY.md 8(b) "API surface to keep") against ours, by ``ast`` over the reference
source where /root/reference is mounted: node ids, widget names / order / defaults / ranges / combo options
(src/interfaces/*.py) and the inference_cli.py flag set with defaults, types, choices and actions
(inference_cli.py:1346-1481).  Plus offline checks that the shim builds its schemas and routes without ComfyUI."""
import ast
import importlib
import os
import sys

import pytest
import torch

from conftest import sub, ROOT
from oracle import reference_loader as rl

needs_ref = pytest.mark.skipif(not rl.available(), reason="/root/reference not mounted")
REF = rl.REFERENCE_ROOT


def _lit(node, env):
    """Literal value of an AST node; names resolve through ``env`` (module constants / local lists); f-strings and
    calls we cannot evaluate become the marker '<dynamic>'."""
    try:
        return ast.literal_eval(node)
    except Exception:
        pass
    if isinstance(node, ast.Name) and node.id in env:
        return env[node.id]
    if isinstance(node, ast.BinOp):
        try:
            return eval(compile(ast.Expression(node), "<ast>", "eval"), {}, {})
        except Exception:
            return "<dynamic>"
    if isinstance(node, ast.Subscript) and isinstance(node.value, ast.Name) and node.value.id in env:
        try:
            return env[node.value.id][ast.literal_eval(node.slice)]
        except Exception:
            return "<dynamic>"
    return "<dynamic>"


def reference_widgets(rel_path, env):
    """[(name, kind, {default, min, max, step, options, optional})] in source order from a define_schema()."""
    tree = ast.parse(open(os.path.join(REF, rel_path)).read())
    out = []
    for node in ast.walk(tree):
        if not (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "Input"):
            continue
        base = node.func.value
        if isinstance(base, ast.Attribute):                       # io.Int.Input(...)
            kind = base.attr
        elif isinstance(base, ast.Call) and getattr(base.func, "attr", "") == "Custom":   # io.Custom("X").Input(...)
            kind = "Custom:" + ast.literal_eval(base.args[0])
        else:
            continue
        name = ast.literal_eval(node.args[0])
        kw = {k.arg: _lit(k.value, env) for k in node.keywords if k.arg in ("default", "min", "max", "step", "options", "optional")}
        out.append((node.lineno, name, kind, kw))
    return [(n, k, kw) for _, n, k, kw in sorted(out)]


def _compare(ours, ref, dynamic_ok=()):
    assert [w.name for w in ours] == [n for n, _, _ in ref]
    for w, (name, kind, kw) in zip(ours, ref):
        assert w.kind == kind, (name, w.kind, kind)
        assert bool(w.optional) == bool(kw.get("optional", False)), name
        for key in ("default", "min", "max", "step", "options"):
            want = kw.get(key)
            if want == "<dynamic>" or name in dynamic_ok and key in ("default", "options"):
                continue
            assert getattr(w, key) == want, (name, key, getattr(w, key), want)


@needs_ref
def test_node_widgets_equal_reference():
    itf = sub("interfaces")
    reg = ast.parse(open(os.path.join(REF, "src/utils/model_registry.py")).read())
    env = {}
    for node in reg.body:                                         # DEFAULT_DIT / DEFAULT_VAE string constants
        if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Name) and isinstance(node.value, ast.Constant):
            env[node.targets[0].id] = node.value.value
    registry = next(n for n in reg.body if isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") == "MODEL_REGISTRY")
    names = [ast.literal_eval(k) for k in registry.value.keys]
    cats = [next((ast.literal_eval(kw.value) for kw in v.keywords if kw.arg == "category"), "dit") for v in registry.value.values]
    assert itf.DIT_MODELS == [n for n, c in zip(names, cats) if c == "dit"]
    assert itf.VAE_MODELS == [n for n, c in zip(names, cats) if c == "vae"]
    assert (itf.DEFAULT_DIT, itf.DEFAULT_VAE) == (env["DEFAULT_DIT"], env["DEFAULT_VAE"])
    env.update(dit_models=itf.DIT_MODELS, vae_models=itf.VAE_MODELS)
    dyn = ("device", "offload_device")                            # device lists depend on the host
    _compare(itf.upscaler_widgets(), reference_widgets("src/interfaces/video_upscaler.py", env), dyn)
    _compare(itf.dit_loader_widgets(), reference_widgets("src/interfaces/dit_model_loader.py", env), dyn)
    _compare(itf.vae_loader_widgets(), reference_widgets("src/interfaces/vae_model_loader.py", env), dyn)
    _compare(itf.compile_widgets(), reference_widgets("src/interfaces/torch_compile_settings.py", env), dyn)
    # node ids and the registry / entry point
    init = open(os.path.join(REF, "src/interfaces/__init__.py")).read()
    for node_id in itf.NODE_TABLE:
        assert node_id in init and hasattr(itf, node_id)
    for rel, node_id in (("video_upscaler.py", "SeedVR2VideoUpscaler"), ("dit_model_loader.py", "SeedVR2LoadDiTModel"),
                         ("vae_model_loader.py", "SeedVR2LoadVAEModel"), ("torch_compile_settings.py", "SeedVR2TorchCompileSettings")):
        src = open(os.path.join(REF, "src/interfaces", rel)).read()
        assert f'node_id="{node_id}"' in src and 'category="SEEDVR2"' in src
    # execute() signatures: same parameter names and defaults
    for rel, cls in (("video_upscaler.py", itf.SeedVR2VideoUpscaler), ("dit_model_loader.py", itf.SeedVR2LoadDiTModel),
                     ("vae_model_loader.py", itf.SeedVR2LoadVAEModel), ("torch_compile_settings.py", itf.SeedVR2TorchCompileSettings)):
        tree = ast.parse(open(os.path.join(REF, "src/interfaces", rel)).read())
        fn = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "execute")
        args = [a.arg for a in fn.args.args][1:]
        defaults = [ast.literal_eval(d) for d in fn.args.defaults]
        import inspect
        sig = inspect.signature(cls.execute)
        assert list(sig.parameters) == args, (rel, list(sig.parameters), args)
        ours_defaults = [p.default for p in sig.parameters.values() if p.default is not inspect.Parameter.empty]
        assert ours_defaults == defaults, (rel, ours_defaults, defaults)


def reference_cli_flags():
    tree = ast.parse(open(os.path.join(REF, "inference_cli.py")).read())
    fn = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "parse_arguments")
    flags = {}
    for node in ast.walk(fn):
        if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "add_argument":
            names = [ast.literal_eval(a) for a in node.args]
            kw = {}
            for k in node.keywords:
                if k.arg in ("default", "choices", "action", "dest"):
                    kw[k.arg] = _lit(k.value, {})
                elif k.arg == "type":
                    kw["type"] = k.value.id
            flags[names[0]] = kw
    return flags


@needs_ref
def test_cli_flags_equal_reference():
    sys.path.insert(0, ROOT)
    cli = importlib.import_module("inference_cli")
    parser = cli.build_parser()
    ours = {}
    for a in parser._actions:
        if a.dest == "help":
            continue
        key = a.option_strings[0] if a.option_strings else a.dest
        ours[key] = a
    ref = reference_cli_flags()
    assert set(ours) == set(ref), (sorted(set(ref) - set(ours)), sorted(set(ours) - set(ref)))
    for key, kw in ref.items():
        a = ours[key]
        if kw.get("action") == "store_true":
            assert a.const is True and a.default is False, key
        else:
            if kw.get("default") != "<dynamic>":
                assert a.default == kw.get("default"), (key, a.default, kw.get("default"))
            assert (a.type.__name__ if a.type else "str") == kw.get("type", "str"), key
        if "choices" in kw and kw["choices"] != "<dynamic>":
            assert list(a.choices) == list(kw["choices"]), key
        if "dest" in kw:
            assert a.dest == kw["dest"], key
    assert parser.allow_abbrev is False


def test_schemas_build_and_loaders_route_without_comfyui():
    itf = sub("interfaces")
    for node_id in itf.NODE_TABLE:
        schema = getattr(itf, node_id).define_schema()
        assert schema.node_id == node_id and schema.category == "SEEDVR2" and len(schema.outputs) == 1
    (cfg,) = itf.SeedVR2LoadDiTModel.execute("seedvr2_ema_3b_fp16.safetensors", "cuda:0", attention_mode="flash_attn_2")
    assert cfg["model"].endswith("fp16.safetensors") and cfg["blocks_to_swap"] == 0
    with pytest.raises(ValueError):
        itf.SeedVR2LoadDiTModel.execute("x.safetensors", "cuda:0", cache_model=True)
    (vcfg,) = itf.SeedVR2LoadVAEModel.execute("ema_vae_fp16.safetensors", "cuda:0", encode_tiled=True, encode_tile_size=1024,
                                             encode_tile_overlap=128, decode_tiled=True, decode_tile_size=1024, decode_tile_overlap=128)
    assert vcfg["encode_tile_size"] == 1024 and vcfg["decode_tile_overlap"] == 128
    with pytest.raises(ValueError):
        itf.SeedVR2LoadVAEModel.execute("ema_vae_fp16.safetensors", "cuda:0", decode_tiled=True, decode_tile_size=64, decode_tile_overlap=64)
    (tc,) = itf.SeedVR2TorchCompileSettings.execute("inductor", "default", False, False, 64, 128)
    assert tc["backend"] == "inductor" and tc["dynamo_recompile_limit"] == 128
    with pytest.raises(FileNotFoundError):
        itf.resolve_model("seedvr2_ema_3b_fp16.safetensors", "/nonexistent")
    with pytest.raises(NotImplementedError):
        itf.SeedVR2VideoUpscaler.execute(torch.zeros(1, 8, 8, 4), cfg, vcfg, 42)


def test_engines_answer_module_style_probes():
    """generation_phases.py probes runner.dit / runner.vae like nn.Modules (next(model.parameters()).device / .dtype at
    :298, :620, :708-712): the engines answer the same questions."""
    from ops_reference import TorchOps
    config, weights, dit, vae = sub("config"), sub("weights"), sub("dit"), sub("vae")
    ops = TorchOps("cpu", act_dtype=torch.float32)
    d = dit.NaDiTEngine(config.DIT_TINY, weights.synth_dit_state_dict(config.DIT_TINY), ops)
    v = vae.VideoVAEEngine(config.VAE_TINY, weights.synth_vae_state_dict(config.VAE_TINY), ops)
    for m in (d, v):
        p = next(m.parameters())
        assert p.device.type == "cpu" and p.dtype in (torch.float32, torch.bfloat16)
        assert m.eval() is m and m.to("cpu") is m and m.requires_grad_(False) is m
        assert sum(1 for _ in m.parameters()) > 10


def test_cli_directory_input_is_one_job_per_media_file(tmp_path):
    """A folder is processed file by file (images and videos, sorted), as the reference's CLI does -- never concatenated into
    one clip; a file is one job."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("svr_cli", os.path.join(ROOT, "inference_cli.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    for name in ("b.png", "a.jpg", "clip.mp4", "notes.txt", "t.pt"):
        (tmp_path / name).write_bytes(b"x")
    jobs = cli.list_inputs(str(tmp_path))
    assert [os.path.basename(j) for j in jobs] == ["a.jpg", "b.png", "clip.mp4", "t.pt"]
    assert cli.list_inputs(str(tmp_path / "b.png")) == [str(tmp_path / "b.png")]
    with pytest.raises(ValueError):
        cli.load_frames(str(tmp_path))
    assert cli.free_port() > 0
