"""Captures the default field values of the reference's HF config classes (loaded by file path,
in the build container only; they import nothing but transformers) into config_defaults.json."""
import importlib.util
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
FILES = {
    "HymetaConfig": ("vllm_hymeta/models/configuration_hymeta.py", "HymetaConfig"),
    "GLAswaConfig": ("vllm_hymeta/model_for_7B/configuration_gla_swa.py", "GLAswaConfig"),
}
KEYS = ["vocab_size", "hidden_size", "num_hidden_layers", "num_attention_heads", "num_key_value_heads",
        "head_dim", "intermediate_size", "max_position_embeddings", "sliding_window", "norm_eps", "rope_theta",
        "num_meta_tokens", "full_attn_layers", "dense_mlp_layers", "interleaved_sliding_window",
        "attn_type_list", "num_layer_experts", "num_local_experts", "num_experts_per_topk",
        "shared_intermediate_size", "use_lower_bound", "gate_logit_normalizer", "gate_low_rank_dim",
        "attn_layers", "model_type", "hidden_act", "bos_token_id", "eos_token_id", "tie_word_embeddings"]

out = {}
for name, (path, cls) in FILES.items():
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cfg = getattr(mod, cls)()
    out[name] = {k: getattr(cfg, k) for k in KEYS if hasattr(cfg, k)}
json.dump(out, open(os.path.join(HERE, "config_defaults.json"), "w"), indent=1, sort_keys=True)
print({k: len(v) for k, v in out.items()})
