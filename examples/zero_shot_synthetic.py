"""Zero-shot scoring of one image against a few prompts -- the flow of the reference's code/test_clip.py
(load -> preprocess -> tokenize -> encode_image / encode_text -> model(image, text) -> softmax), on synthetic
inputs: there is no CLIP.png, BPE table or pretrained checkpoint offline, so the image is seeded noise, the
prompts are token-id rows and the weights are the seeded generator's (pass --weights <state dict file> and
--bpe <bpe_simple_vocab_16e6.txt.gz> to run it for real).

    python examples/zero_shot_synthetic.py [--model ViT-B/32] [--weights ckpt] [--bpe vocab.gz]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mmr_amd as clip  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="ViT-B/32")
    ap.add_argument("--weights", default="synthetic", help="checkpoint file (OpenAI / HF / this package naming), or \"synthetic\" (default here: seeded random weights)")
    ap.add_argument("--bpe", default=None, help="CLIP BPE merge table; without it prompts are synthetic token ids")
    args = ap.parse_args(argv)

    device = "cuda"
    model, preprocess = clip.load(args.model, device=device, weights=args.weights)

    rng = np.random.default_rng(0)
    picture = rng.integers(0, 256, size=(360, 480, 3), dtype=np.uint8)          # stands in for Image.open("CLIP.png")
    image = preprocess(picture).unsqueeze(0).to(device)
    if args.bpe:
        text = clip.tokenize(["a diagram", "a dog", "a cat"], bpe_path=args.bpe).to(device)
    else:
        text = clip.synth.synth_token_ids(3, model.context_length, model.vocab_size, seed=7).to(device)

    with torch.no_grad():
        image_features = model.encode_image(image)
        text_features = model.encode_text(text)
        logits_per_image, logits_per_text = model(image, text)
        probs = logits_per_image.softmax(dim=-1).cpu().numpy()

    print("image_features", tuple(image_features.shape), "text_features", tuple(text_features.shape))
    print("Label probs:", probs)
    return probs


if __name__ == "__main__":
    main()
