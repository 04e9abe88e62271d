"""Hyper-parameters of the RecNet hot path.

Attribute names (and defaults) follow the reference's `config.TrainConfig` (config.py:27-93) so code
written against `C.<name>` keeps working; only the attributes the train-step path reads are kept
(data-loader paths, tensorboard tags and the run-id string builder are out of scope).
Two additions: `precision` and the data-parallel fields.
"""

_DEFAULTS = dict(
    # model family (config.py:28-33).  NB the reference's literal default is decoder_model="GRU";
    # every published run and the benchmark use LSTM (SURVEY.md §8a trap 14), and the HIP path
    # implements the LSTM cell.
    model="RecNet", decoder_model="LSTM", reconstructor_model="LSTM", device="cuda",
    # vocabulary / caption shape (config.py:48-56)
    caption_max_len=30, batch_size=100, init_word2idx={"<PAD>": 0, "<SOS>": 1, "<EOS>": 2},
    # embedding (config.py:57-59)
    embedding_size=468, embedding_dropout=0.5, embedding_scale=1,
    # encoder features (config.py:62-63)
    encoder_output_size=1536, encoder_output_len=28,
    # decoder (config.py:66-71)
    decoder_n_layers=1, decoder_hidden_size=512, decoder_attn_size=128, decoder_dropout=0.5,
    decoder_out_dropout=0.5, decoder_teacher_forcing_ratio=1.0,
    # reconstructor (config.py:74-82)
    use_recon=True, reconstructor_type="local", reconstructor_n_layers=1, reconstructor_hidden_size=1536,
    reconstructor_decoder_dropout=0.5, reconstructor_dropout=0.5, reconstructor_attn_size=128,
    # optimisation (config.py:85-93)
    n_iterations=100000, decoder_learning_rate=1e-5, reconstructor_learning_rate=1e-6,
    decoder_weight_decay=1e-5, reconstructor_weight_decay=1e-5, decoder_use_amsgrad=True,
    reconstructor_use_amsgrad=False, use_gradient_clip=True, gradient_clip=50.0,
    # constants the reference creates inline (train.py:151,188,225)
    decoder_lambda_reg=1e-3, reconstructor_lambda_reg=1e-2, lambda_recon=1.0,
    # additions
    precision="bf16",            # "bf16" (bf16 MFMA operands, fp32 accumulate) | "f32" (exact fp32 MFMA)
    decoder_attn_normalize="none",   # "none": the reference (decoder.py:30's softmax is never called) | "softmax": opt-in
    dropout_seed=42,
)


class TrainConfig:
    """Class-attribute bag like the reference's (`from config import TrainConfig as C`)."""


for _k, _v in _DEFAULTS.items():
    setattr(TrainConfig, _k, _v)


def make_config(**overrides):
    """A fresh config class (so tests do not mutate the shared TrainConfig)."""
    unknown = set(overrides) - set(_DEFAULTS)
    if unknown:
        raise AttributeError("unknown config attribute(s): %s" % sorted(unknown))
    return type("TrainConfig", (TrainConfig,), dict(overrides))
