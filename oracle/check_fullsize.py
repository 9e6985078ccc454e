"""Build-container only: oracle vs the real reference at full size
(12-layer ViT-B/14, 518x518, B=1, synthetic weights).  Prints the max relative
error (1e-5 class: fp32 summation order only)."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.make_goldens import load_reference, quiet
from oracle.weights import reference_state_spec, synth_state_dict
from oracle import cogaim_oracle as O

def main():
    from transformers import Dinov2Config, Dinov2Model
    ref = load_reference()
    ref.Dinov2Model.from_pretrained = classmethod(
        lambda cls, name, *a, **k: Dinov2Model(Dinov2Config(image_size=518)))
    cfg = {"cognitive_modules": ["ambient_stream", "iterative_focal_stream", "exif_prior_database"]}
    with quiet():
        model = ref.create_model(cfg, {"num_cameras": 71}).eval()
    spec = reference_state_spec()
    assert list(model.state_dict().keys()) == list(spec.keys())
    sd = synth_state_dict(spec, seed=11)
    model.load_state_dict(sd)
    torch.manual_seed(5)
    x = torch.randn(1, 3, 518, 518)
    exif = {"camera_idx": torch.tensor([3]), "focal_length": torch.tensor([50.0]),
            "aperture": torch.tensor([2.8]), "iso": torch.tensor([100.0])}
    t = time.time()
    with torch.no_grad(), quiet():
        d, c, a = model(x, exif, return_attention=True)
        _, _, ga = model.forward_with_guidance(x, exif, "top-left", return_attention=True)
    t_ref = time.time() - t
    t = time.time()
    with torch.no_grad():
        d2, c2, a2 = O.forward(sd, x, exif, return_attention=True)
        _, _, ga2 = O.forward_with_guidance(sd, x, exif, "top-left")
    t_or = time.time() - t
    rel = lambda p, q: float(((p - q).abs() / q.abs().clamp_min(1e-12)).max())
    print(f"depth ref {d.item():.6f} oracle {d2.item():.6f} rel {rel(d2, d):.2e}")
    print(f"conf  ref {c.item():.6f} oracle {c2.item():.6f} rel {rel(c2, c):.2e}")
    print(f"attn max rel {rel(a2, a):.2e}; guided attn max abs {float((ga2-ga).abs().max()):.2e} (peak {float(ga.max()):.3e})")
    print(f"reference {t_ref:.1f}s  oracle {t_or:.1f}s  threads {torch.get_num_threads()}")

if __name__ == "__main__":
    main()
