"""Global epistasis surrogate -- same constructor as
flexs/baselines/models/global_epistasis_model.py:15-48."""
from . import keras_model


class GlobalEpistasisModel(keras_model.KerasModel):
    """Flatten -> Dense(1,relu) -> Dense(H,relu) -> Dense(H,relu) -> Dense(1)
    (global_epistasis_model.py:26-36)."""

    def __init__(self, seq_len: int, hidden_size: int, alphabet: str, loss="MSE", name: str = None,
                 batch_size: int = 256, epochs: int = 20, device: int = None, seed: int = None):
        model = keras_model.Architecture("ge", seq_len, len(alphabet), hidden_size, loss=loss, seed=seed)
        if name is None:
            # the reference reuses the MLP name here (global_epistasis_model.py:39-40, sic);
            # run logs carry it, so it is kept
            name = f"MLP_hidden_size_{hidden_size}"
        super().__init__(model, alphabet=alphabet, name=name, batch_size=batch_size, epochs=epochs, device=device)
