"""Few-shot image search over a gallery -- the flow of the reference's code/search_image.py main():
class text embeddings -> build_cache over the dataset -> per class: 10 reference images -> outlier-filtered
mean -> averaged with the class text embedding -> similarity of every gallery image to that query ->
positives vs negatives.  Synthetic dataset: each class is a seeded prototype image plus per-image noise, so
images of a class are close in pixel space (and, the encoder being continuous, in feature space) even with
random-init weights.  Adds what the reference does not have: a fused top-k over the same gallery.

    python examples/gallery_search_synthetic.py [--classes 6] [--per-class 200] [--model ViT-B/32]
"""
import argparse
import os
import random
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mmr_amd as clip  # noqa: E402
from mmr_amd import gallery  # noqa: E402


def make_dataset(n_classes, per_class, size, seed=0):
    """{relpath: uint8 HWC image}, labels; class c = prototype_c + noise."""
    rng = np.random.default_rng(seed)
    protos = rng.integers(0, 256, size=(n_classes, size, size, 3)).astype(np.float32)
    images, labels = {}, {}
    for c in range(n_classes):
        for i in range(per_class):
            img = np.clip(protos[c] + rng.normal(0, 40, size=protos[c].shape), 0, 255).astype(np.uint8)
            key = f"class{c}/img{i:04d}.jpg"
            images[key], labels[key] = img, c
    return images, labels


def outlier_filter(model, preprocess, imgs):
    """Robust centre of a few reference images: mean of the features within the 95th percentile of cosine
    distance to the plain mean (not re-normalised, like the reference)."""
    with torch.no_grad():
        x = torch.stack([preprocess(im) for im in imgs], dim=0)
        f = model.encode_image(x)
        f /= f.norm(dim=-1, keepdim=True)
        f = f.cpu().numpy()
    center = f.mean(axis=0)
    dist = 1 - f @ center
    keep = dist <= np.percentile(dist, 95)
    return torch.tensor(f[keep].mean(axis=0)).cuda()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="ViT-B/32")
    ap.add_argument("--classes", type=int, default=6)
    ap.add_argument("--per-class", type=int, default=200)
    ap.add_argument("--image-size", type=int, default=256)
    ap.add_argument("--weights", default="synthetic", help="checkpoint file, or \"synthetic\" (default here: seeded random weights)")
    args = ap.parse_args(argv)
    random.seed(1)
    torch.manual_seed(1)

    model, preprocess = clip.load(args.model, device="cuda", weights=args.weights)
    model.eval()
    images, labels = make_dataset(args.classes, args.per_class, args.image_size)
    keys = list(images)

    texts = clip.synth.synth_token_ids(args.classes, model.context_length, model.vocab_size, seed=3).cuda()   # clip.tokenize(class_names)
    class_embeddings = model.encode_text(texts)
    class_embeddings /= class_embeddings.norm(dim=-1, keepdim=True)

    with tempfile.TemporaryDirectory() as tmp:
        cache = os.path.join(tmp, "caches", "search", "features.pkl")
        keys, features = gallery.build_cache(model, keys, lambda k: preprocess(images[k]), cache_path=cache, batch_size=256)
        again_keys, again = gallery.build_cache(model, keys, None, cache_path=cache)       # second call: served from the pickle
        assert again_keys == keys and torch.equal(again.float().cpu(), features.float().cpu())
    targets = np.array([labels[k] for k in keys])
    print(f"gallery: {features.shape[0]} images x {features.shape[1]} dims ({features.dtype})")

    report = []
    for c in range(args.classes):
        members = [k for k in keys if labels[k] == c]
        refs = random.sample(members, 10)
        robust = outlier_filter(model, preprocess, [images[k] for k in refs])
        # the reference's two query variants: the filtered image mean alone, and averaged with the class text
        # embedding.  With random-init weights text and image embeddings are not aligned, so only the first is
        # expected to separate the synthetic classes; both run the same kernels.
        for name, query in (("image mean", robust), ("image+text", (robust + class_embeddings[c]) / 2.)):
            sim = clip.similarity(features, query, 100.).cpu().numpy()            # 100. * features @ ref.t()
            pos, neg = sim[targets == c], sim[targets != c]
            vals, idx = clip.cosine_topk(query[None, :], features, k=10, scale=100.)
            hits = int((targets[idx[0].cpu().numpy()] == c).sum())
            assert np.allclose(vals[0].cpu().numpy(), np.sort(sim)[::-1][:10], atol=1e-3 * 100)   # top-k == sorted similarity
            report.append((c, name, float(pos.mean()), float(neg.mean()), hits))
            print(f"class {c} [{name:10s}]: mean score positives {pos.mean():7.3f}  negatives {neg.mean():7.3f}  "
                  f"top-10 of class: {hits}/10")
    return report


if __name__ == "__main__":
    main()
