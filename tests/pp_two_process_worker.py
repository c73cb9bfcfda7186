"""Worker of tests/test_gpu_pipeline_processes.py: one pipeline stage of the 70B-family model per process, both on GPU 0,
the stages' hand-offs over a gloo group (host copies).  argv: rank, world, port, result file."""
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [HERE, ROOT, os.path.join(ROOT, "vllm-hymeta_amd")]


def main():
    rank, world, port, result = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from _util import gen
    from test_gpu_layers import _md
    from test_gpu_model_70b import D, _checkpoint
    from vllm_hymeta.models.configuration_hymeta import HymetaConfig
    from vllm_hymeta.models.modeling_hymeta import HymetaForCausalLM
    from vllm_hymeta.parallel import recv_from_stage, send_to_stage

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    g_ = gen(51)
    cfg = HymetaConfig(vocab_size=512, hidden_size=512, num_hidden_layers=4, num_attention_heads=4, num_key_value_heads=2,
                       intermediate_size=512, shared_intermediate_size=256, num_local_experts=4, num_meta_tokens=32,
                       sliding_window=48, max_position_embeddings=1024)
    ckpt = _checkpoint(g_, cfg)
    stage = HymetaForCausalLM(cfg, max_num_seqs=4, device=dev, pp_rank=rank, pp_size=world, pp_group=dist.group.WORLD)
    stage.load_weights(ckpt.items())          # the meta rows travel over the group here
    assert stage.model.has_meta_cache
    lens = [50, 31]
    ids = torch.cat([torch.randint(0, cfg.vocab_size, (L,), generator=g_) for L in lens]).to(dev)
    pos = torch.cat([torch.arange(L) for L in lens]).to(dev)
    bs = 16
    per_seq = (max(lens) + bs - 1) // bs
    slots = torch.tensor([(s * per_seq + t // bs) * bs + t % bs for s, L in enumerate(lens) for t in range(L)], device=dev)
    md = _md(dev, prefill_lens=lens, context_lens=[0, 0], slot_mapping=slots)
    kv = [torch.zeros(2, 2 * per_seq, bs, cfg.num_key_value_heads, D, dtype=torch.bfloat16, device=dev) for _ in stage.model.attn_layers]
    run = dict(request_ids_to_seq_ids={"a": [0], "b": [1]}, finished_requests_ids=[])
    x = None
    if rank > 0:
        like = torch.empty(sum(lens), cfg.hidden_size, dtype=torch.bfloat16, device=dev)
        x = stage.make_empty_intermediate_tensors(sum(lens), torch.bfloat16, dev)
        x["hidden_states"] = recv_from_stage(like, dist.group.WORLD, rank - 1)
    out = stage(ids if rank == 0 else None, pos, intermediate_tensors=x, attn_metadata=md, kv_caches=kv, **run)
    if rank < world - 1:
        send_to_stage(out["hidden_states"], dist.group.WORLD, rank + 1)
        verdict = "sent"
    else:
        whole = HymetaForCausalLM(cfg, max_num_seqs=4, device=dev)
        whole.load_weights(ckpt.items())
        kvw = [torch.zeros(2, 2 * per_seq, bs, cfg.num_key_value_heads, D, dtype=torch.bfloat16, device=dev) for _ in whole.model.attn_layers]
        ref = whole(ids, pos, attn_metadata=md, kv_caches=kvw, **run)
        torch.cuda.synchronize()
        verdict = "equal" if torch.equal(ref, out) else f"differs in {int((ref != out).sum())} values"
    torch.cuda.synchronize()
    dist.barrier()
    with open(result, "w") as f:
        f.write(verdict)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
