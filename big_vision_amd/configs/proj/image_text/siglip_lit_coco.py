"""LiT-style SigLIP fine-tune on image-caption pairs: frozen image tower, trainable text tower.

Counterpart of big_vision/configs/proj/image_text/siglip_lit_coco.py (historical name
`lit_coco.py`, README.md:83) for the accelerated hot path: the model / optimizer / schedule
hyper-parameters are the reference's (:79-104: ViT-B/16 with the cls token, head_zeroinit False,
temperature 10, bias -2.71, out_dim (None, 768); lr 1e-3, wd 1e-2, cosine schedule with
max(3 % of the run, 100) warm-up steps, image tower frozen through the schedule, clip norm 1),
so a run configured from either file takes the same step.  Input pipeline, tokenizer,
checkpoint locations and the retrieval evaluator's dataset plumbing of the reference file are
outside the hot path: this file states the tensors the step consumes instead (`init_shapes`,
`input.batch_size`).  `txt`: 'bert_base' (default, as in the reference: models.proj.flaxformer.bert,
Adam), 'bert_large' (Adafactor, :91-94) or 'transformer_b' (the in-repo text transformer of width 768).
"""
import big_vision.configs.common as bvcc
from ml_collections import ConfigDict

_IMG = {"B/16": ("B/16", 768), "L/16": ("L/16", 1024)}


def get_config(arg=None):
  arg = bvcc.parse_arg(arg, res=224, runlocal=False, token_len=16, txt="bert_base", img="B/16",
                       init="", img_head=False, batch_size=512)
  variant, dim = _IMG[arg.img]
  c = ConfigDict()
  c.input = dict(batch_size=arg.batch_size if not arg.runlocal else 32)
  c.total_steps = 5_000 if not arg.runlocal else 1
  c.init_shapes = [(1, arg.res, arg.res, 3), (1, arg.token_len)]
  c.init_types = ["float32", "int32"]
  c.log_training_steps = 50
  c.ckpt_steps = 1000

  c.model_name = "proj.image_text.two_towers"
  c.model_load = {}
  if arg.init:
    c.model_init = arg.init
  c.model = ConfigDict()
  c.model.image_model = "vit"
  c.model.image = ConfigDict(dict(variant=variant, pool_type="tok", head_zeroinit=False))
  if arg.txt in ("bert_base", "bert_large"):
    txt_name = arg.txt[len("bert_"):]
    c.model.text_model = "proj.flaxformer.bert"
    c.model.text = ConfigDict(dict(config=txt_name, head_zeroinit=False))
    c.optax_name = "scale_by_adam" if txt_name == "base" else "big_vision.scale_by_adafactor"   # :91-94
  else:
    c.model.text_model = "proj.image_text.text_transformer"
    c.model.text = ConfigDict(dict(variant="B", vocab_size=32_000))
    c.optax_name = "scale_by_adam"
  c.model.temperature_init = 10.0
  c.model.out_dim = (dim if arg.img_head else None, dim)
  c.model.bias_init = -2.71

  c.lr = 0.001
  c.wd = 0.01
  warmup = max(int(0.03 * c.total_steps), 100)
  c.schedule = [("img/.*", None),                                     # freezes the image tower
                (".*", dict(decay_type="cosine", warmup_steps=warmup))]
  c.grad_clip_norm = 1.0
  c.sharding_strategy = [(".*", "replicate")]
  return c
