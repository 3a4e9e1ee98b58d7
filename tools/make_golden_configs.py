#!/usr/bin/env python3
"""tests/golden/configs.json: every configuration file the reference ships (script/config/**/*.txt) run through the
reference's OWN argument parser (script/models/options.py: config_parser), reduced to the arguments the render path reads
(create_nerf, nerfh_nff.py:628-736; render_kwargs; the refinement loop's tinyscale / learning rates).  Data only: the GPU test
(tests/test_gpu_surface.py::test_every_shipped_config_runs_on_the_hip_path) builds the networks from these namespaces with
the drop-in create_nerf and renders a small frame with each distinct setting.

`configargparse` is not installed here; the shim below gives argparse the two things the reference uses from it
(`is_config_file` arguments and `key = value` config files with `#` comments).  CPU only, this container only."""
import argparse
import glob
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/script"


class ConfigArgParser(argparse.ArgumentParser):
    def __init__(self, *a, **k):
        k.pop("config_file_parser_class", None)
        k.pop("default_config_files", None)
        super().__init__(*a, **k)
        self._config_dests = []

    def add_argument(self, *a, **k):
        is_cfg = k.pop("is_config_file", False)
        act = super().add_argument(*a, **k)
        if is_cfg:
            self._config_dests.append(act.dest)
        return act

    def parse_file(self, path):
        argv = []
        for line in open(path):
            line = line.split("#", 1)[0].strip()
            if not line:
                continue
            key, _, val = line.partition("=")
            key, val = key.strip(), val.strip()
            act = next((x for x in self._actions if x.dest == key or ("--" + key) in x.option_strings), None)
            if act is None:
                raise SystemExit(f"{path}: unknown option {key}")
            if isinstance(act, (argparse._StoreTrueAction, argparse._StoreFalseAction)):
                if val.lower() in ("true", "1", "yes", ""):
                    argv.append("--" + key)
            elif act.nargs in ("+", "*"):
                argv += ["--" + key] + val.replace("[", " ").replace("]", " ").replace(",", " ").split()
            else:
                argv += ["--" + key, val]
        return self.parse_args(argv)


def main():
    sys.modules["configargparse"] = types.SimpleNamespace(ArgumentParser=ConfigArgParser, YAMLConfigFileParser=None)
    sys.path.insert(0, REF)
    import importlib.util

    def load(rel):
        spec = importlib.util.spec_from_file_location("ref_" + rel.replace("/", "_"), os.path.join(REF, rel))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m
    # run_nefes.py parses with models/options.py (stage1 / stage2 files), test_refinement.py with dm/options.py (the *_DFM files)
    parsers = {"models/options.py": load("models/options.py"), "dm/options.py": load("dm/options.py")}
    keep = ["netdepth", "netwidth", "multires", "multires_views", "i_embed", "reduce_embedding", "use_viewdirs", "N_samples",
            "N_importance", "perturb", "raw_noise_std", "white_bkgd", "lindisp", "no_ndc", "dataset_type", "netchunk", "chunk",
            "lrate", "nerfh_nff", "nerfh_nff2", "NeRFW", "encode_hist", "transient_at_test", "use_fine_only", "tinyscale", "lr_r", "lr_t",
            "opt_iter", "in_channels_a", "in_channels_t", "use_fusion_res", "no_fusion_BN", "new_schedule", "semantic",
            "feature_matching_lvl", "pose_only", "tcnn", "no_grad_update"]
    out = {}
    for path in sorted(glob.glob(os.path.join(REF, "config", "**", "*.txt"), recursive=True)):
        which = "dm/options.py" if path.endswith("_DFM.txt") else "models/options.py"
        parser = parsers[which].config_parser()
        args = parser.parse_file(path)
        rel = os.path.relpath(path, REF)
        out[rel] = dict({k: getattr(args, k) for k in keep if hasattr(args, k)}, parser=which)
    dst = os.path.join(ROOT, "tests", "golden", "configs.json")
    json.dump(out, open(dst, "w"), indent=0, sort_keys=True)
    distinct = {json.dumps(v, sort_keys=True) for v in out.values()}
    print(f"{len(out)} configuration files, {len(distinct)} distinct render-path settings -> {dst}")
    for d in sorted(distinct):
        print("  ", d[:400])


if __name__ == "__main__":
    main()
