"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the BERT-style text classifier (Taiyi Chinese text tower).

Plain-torch fp32 restatement of ``BertForSequenceClassification(input_ids).logits`` as the reference calls it
(code/test_taiyi.py:12,24: ids only -- no attention_mask, no token_type_ids -- so every token attends to every
token, pads included, and token type 0 is used throughout) and of the ``text_encoder(**tokenizer_output)`` form
(CLIP/union_dataset.py:312-314: attention_mask = additive large-negative bias on padded KEYS,
modeling_bert.py / modeling_attn_mask_utils; token_type_ids index the type embedding).  Follows
transformers/models/bert/modeling_bert.py @5.15.0: embeddings :53-108, self-attention :111-204,
self-output :282-293, intermediate/output :325-351, pooler :451-464, classifier in
BertForSequenceClassification.  Pinned by tests/golden/bert_*.npz (oracle/make_golden.py --only bert).
Only tests/ may import this.
"""
import torch
import torch.nn.functional as F


def bert_logits(w, cfg, ids: torch.Tensor, stages: dict = None, attention_mask: torch.Tensor = None,
                token_type_ids: torch.Tensor = None) -> torch.Tensor:
    N, T = ids.shape
    d, heads = cfg.width, cfg.heads
    dh = d // heads
    ids = ids.long()
    types = torch.zeros_like(ids) if token_type_ids is None else token_type_ids.long()
    bias = None
    if attention_mask is not None:
        bias = torch.zeros(N, 1, 1, T)
        bias.masked_fill_(attention_mask.view(N, 1, 1, T) == 0, torch.finfo(torch.float32).min)
    h = (w["b.tok"][ids] + w["b.type"][types]) + w["b.pos"][:T]
    h = F.layer_norm(h, (d,), w["b.ln_emb.w"], w["b.ln_emb.b"], cfg.ln_eps)
    if stages is not None:
        stages["embed"] = h.clone()
    for i in range(cfg.layers):
        p = f"b.l{i}"
        qkv = h @ w[f"{p}.qkv.w"].t() + w[f"{p}.qkv.b"]
        q, k, v = qkv.split(d, dim=-1)
        q = q.view(N, T, heads, dh).transpose(1, 2)
        k = k.view(N, T, heads, dh).transpose(1, 2)
        v = v.view(N, T, heads, dh).transpose(1, 2)
        scores = (q @ k.transpose(-1, -2)) * dh ** -0.5
        att = torch.softmax(scores if bias is None else scores + bias, dim=-1)
        ctx = (att @ v).transpose(1, 2).reshape(N, T, d)
        h = F.layer_norm(ctx @ w[f"{p}.out.w"].t() + w[f"{p}.out.b"] + h, (d,), w[f"{p}.ln1.w"], w[f"{p}.ln1.b"], cfg.ln_eps)
        u = F.gelu(h @ w[f"{p}.fc1.w"].t() + w[f"{p}.fc1.b"])            # exact (erf) GELU
        h = F.layer_norm(u @ w[f"{p}.fc2.w"].t() + w[f"{p}.fc2.b"] + h, (d,), w[f"{p}.ln2.w"], w[f"{p}.ln2.b"], cfg.ln_eps)
        if stages is not None and (i == 0 or i == cfg.layers - 1):
            stages[f"layer{i}"] = h.clone()
    pooled = torch.tanh(h[:, 0] @ w["b.pool.w"].t() + w["b.pool.b"])
    return pooled @ w["b.cls.w"].t() + w["b.cls.b"]
