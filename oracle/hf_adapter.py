"""TEST INFRASTRUCTURE ONLY -- builds the third-party model the reference calls.

Used only by ``oracle/make_golden.py`` in the dev container (transformers is not needed,
and not imported, on the GPU box).  Constructs ``transformers.CLIPModel`` from a *local*
config object (no hub fetch) and loads this repo's synthetic weights into it, so the
committed golden vectors come from the same code the reference executes
(code/test_taiyi.py:17-30: ``CLIPModel...get_image_features``; ``.logit_scale``).
"""
import os

os.environ.setdefault("HF_HUB_OFFLINE", "1")
os.environ.setdefault("TRANSFORMERS_OFFLINE", "1")

import torch


def build_hf_clip(ccfg, w):
    from transformers import CLIPConfig, CLIPModel, CLIPTextConfig, CLIPVisionConfig

    v, t = ccfg.vision, ccfg.text
    vcfg = CLIPVisionConfig(
        hidden_size=v.width, intermediate_size=v.mlp, num_hidden_layers=v.layers,
        num_attention_heads=v.heads, image_size=v.image_size, patch_size=v.patch,
        hidden_act="quick_gelu", layer_norm_eps=v.ln_eps, projection_dim=ccfg.embed_dim,
    )
    tcfg = CLIPTextConfig(
        vocab_size=t.vocab, hidden_size=t.width, intermediate_size=t.mlp,
        num_hidden_layers=t.layers, num_attention_heads=t.heads,
        max_position_embeddings=t.tokens, hidden_act="quick_gelu", layer_norm_eps=t.ln_eps,
        projection_dim=ccfg.embed_dim, eos_token_id=t.vocab - 1, bos_token_id=t.vocab - 2,
        pad_token_id=0,
    )
    cfg = CLIPConfig(text_config=tcfg.to_dict(), vision_config=vcfg.to_dict(),
                     projection_dim=ccfg.embed_dim, logit_scale_init_value=ccfg.logit_scale_init)
    cfg._attn_implementation = "eager"
    model = CLIPModel(cfg).eval()

    sd = {}
    sd["logit_scale"] = w["logit_scale"]
    sd["vision_model.embeddings.class_embedding"] = w["v.cls"]
    sd["vision_model.embeddings.patch_embedding.weight"] = w["v.patch_w"]
    sd["vision_model.embeddings.position_embedding.weight"] = w["v.pos"]
    sd["vision_model.pre_layrnorm.weight"] = w["v.ln_pre.w"]
    sd["vision_model.pre_layrnorm.bias"] = w["v.ln_pre.b"]
    sd["vision_model.post_layernorm.weight"] = w["v.ln_post.w"]
    sd["vision_model.post_layernorm.bias"] = w["v.ln_post.b"]
    sd["visual_projection.weight"] = w["v.proj"]
    sd["text_model.embeddings.token_embedding.weight"] = w["t.tok"]
    sd["text_model.embeddings.position_embedding.weight"] = w["t.pos"]
    sd["text_model.final_layer_norm.weight"] = w["t.ln_final.w"]
    sd["text_model.final_layer_norm.bias"] = w["t.ln_final.b"]
    sd["text_projection.weight"] = w["t.proj"]
    for tower, pre, cfgt in (("vision_model", "v", v), ("text_model", "t", t)):
        d = cfgt.width
        for i in range(cfgt.layers):
            hp = f"{tower}.encoder.layers.{i}"
            p = f"{pre}.l{i}"
            qw, kw, vw = w[f"{p}.qkv.w"].split(d, dim=0)
            qb, kb, vb = w[f"{p}.qkv.b"].split(d, dim=0)
            sd[f"{hp}.self_attn.q_proj.weight"], sd[f"{hp}.self_attn.q_proj.bias"] = qw, qb
            sd[f"{hp}.self_attn.k_proj.weight"], sd[f"{hp}.self_attn.k_proj.bias"] = kw, kb
            sd[f"{hp}.self_attn.v_proj.weight"], sd[f"{hp}.self_attn.v_proj.bias"] = vw, vb
            sd[f"{hp}.self_attn.out_proj.weight"] = w[f"{p}.out.w"]
            sd[f"{hp}.self_attn.out_proj.bias"] = w[f"{p}.out.b"]
            sd[f"{hp}.layer_norm1.weight"] = w[f"{p}.ln1.w"]
            sd[f"{hp}.layer_norm1.bias"] = w[f"{p}.ln1.b"]
            sd[f"{hp}.layer_norm2.weight"] = w[f"{p}.ln2.w"]
            sd[f"{hp}.layer_norm2.bias"] = w[f"{p}.ln2.b"]
            sd[f"{hp}.mlp.fc1.weight"] = w[f"{p}.fc1.w"]
            sd[f"{hp}.mlp.fc1.bias"] = w[f"{p}.fc1.b"]
            sd[f"{hp}.mlp.fc2.weight"] = w[f"{p}.fc2.w"]
            sd[f"{hp}.mlp.fc2.bias"] = w[f"{p}.fc2.b"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    missing = [m for m in missing if "position_ids" not in m]
    if missing or unexpected:
        raise RuntimeError(f"HF state_dict mismatch: missing={missing} unexpected={unexpected}")
    return model


@torch.no_grad()
def hf_image_features(model, pixels):
    out = model.get_image_features(pixel_values=pixels)
    return out.pooler_output if hasattr(out, "pooler_output") else out


@torch.no_grad()
def hf_text_features(model, ids):
    out = model.get_text_features(input_ids=ids)
    return out.pooler_output if hasattr(out, "pooler_output") else out


def build_hf_bert(cfg, w):
    """transformers.BertForSequenceClassification from a LOCAL config + this repo's seeded weights."""
    from transformers import BertConfig, BertForSequenceClassification

    hc = BertConfig(vocab_size=cfg.vocab, hidden_size=cfg.width, num_hidden_layers=cfg.layers,
                    num_attention_heads=cfg.heads, intermediate_size=cfg.mlp, hidden_act="gelu",
                    max_position_embeddings=cfg.max_positions, type_vocab_size=2, layer_norm_eps=cfg.ln_eps,
                    num_labels=cfg.embed_dim, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    hc._attn_implementation = "eager"
    model = BertForSequenceClassification(hc).eval()
    d = cfg.width
    sd = {
        "bert.embeddings.word_embeddings.weight": w["b.tok"],
        "bert.embeddings.position_embeddings.weight": w["b.pos"],
        "bert.embeddings.token_type_embeddings.weight": w["b.type"],
        "bert.embeddings.LayerNorm.weight": w["b.ln_emb.w"],
        "bert.embeddings.LayerNorm.bias": w["b.ln_emb.b"],
        "bert.pooler.dense.weight": w["b.pool.w"], "bert.pooler.dense.bias": w["b.pool.b"],
        "classifier.weight": w["b.cls.w"], "classifier.bias": w["b.cls.b"],
    }
    for i in range(cfg.layers):
        hp, p = f"bert.encoder.layer.{i}", f"b.l{i}"
        qw, kw, vw = w[f"{p}.qkv.w"].split(d, dim=0)
        qb, kb, vb = w[f"{p}.qkv.b"].split(d, dim=0)
        sd[f"{hp}.attention.self.query.weight"], sd[f"{hp}.attention.self.query.bias"] = qw, qb
        sd[f"{hp}.attention.self.key.weight"], sd[f"{hp}.attention.self.key.bias"] = kw, kb
        sd[f"{hp}.attention.self.value.weight"], sd[f"{hp}.attention.self.value.bias"] = vw, vb
        sd[f"{hp}.attention.output.dense.weight"] = w[f"{p}.out.w"]
        sd[f"{hp}.attention.output.dense.bias"] = w[f"{p}.out.b"]
        sd[f"{hp}.attention.output.LayerNorm.weight"] = w[f"{p}.ln1.w"]
        sd[f"{hp}.attention.output.LayerNorm.bias"] = w[f"{p}.ln1.b"]
        sd[f"{hp}.intermediate.dense.weight"] = w[f"{p}.fc1.w"]
        sd[f"{hp}.intermediate.dense.bias"] = w[f"{p}.fc1.b"]
        sd[f"{hp}.output.dense.weight"] = w[f"{p}.fc2.w"]
        sd[f"{hp}.output.dense.bias"] = w[f"{p}.fc2.b"]
        sd[f"{hp}.output.LayerNorm.weight"] = w[f"{p}.ln2.w"]
        sd[f"{hp}.output.LayerNorm.bias"] = w[f"{p}.ln2.b"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    missing = [m for m in missing if "position_ids" not in m and "token_type_ids" not in m]
    if missing or unexpected:
        raise RuntimeError(f"HF BERT state_dict mismatch: missing={missing} unexpected={unexpected}")
    return model
