"""Baseline MLP surrogate -- same constructor as flexs/baselines/models/mlp.py:10-44."""
from . import keras_model


class MLP(keras_model.KerasModel):
    """Flatten -> Dense(H,relu) x3 -> Dense(1)   (mlp.py:21-31)."""

    def __init__(self, seq_len, hidden_size, alphabet, loss="MSE", name=None, batch_size=256, epochs=20,
                 device=None, seed=None):
        model = keras_model.Architecture("mlp", seq_len, len(alphabet), hidden_size, loss=loss, seed=seed)
        if name is None:
            name = f"MLP_hidden_size_{hidden_size}"                                # mlp.py:35-36
        super().__init__(model, alphabet=alphabet, name=name, batch_size=batch_size, epochs=epochs, device=device)
