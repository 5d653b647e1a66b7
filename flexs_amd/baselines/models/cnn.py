"""Baseline CNN surrogate -- same constructor as flexs/baselines/models/cnn.py:10-67."""
from . import keras_model


class CNN(keras_model.KerasModel):
    """Conv1D(F,K,valid,relu) -> Conv1D(F,K,same,relu) -> MaxPool1D(1) ->
    Conv1D(F,len(alphabet)-1,same,relu) -> GlobalMaxPool1D -> Dense(H,relu) x2 ->
    Dropout(.25) -> Dense(1)   (cnn.py:23-54), scored by the fused MFMA kernel."""

    def __init__(
        self,
        seq_len: int,
        num_filters: int,
        hidden_size: int,
        alphabet: str,
        loss="MSE",
        kernel_size: int = 5,
        name: str = None,
        batch_size: int = 256,
        epochs: int = 20,
        device: int = None,
        seed: int = None,
    ):
        model = keras_model.Architecture("cnn", seq_len, len(alphabet), hidden_size, num_filters=num_filters,
                                         kernel_size=kernel_size, loss=loss, seed=seed)
        if name is None:
            name = f"CNN_hidden_size_{hidden_size}_num_filters_{num_filters}"     # cnn.py:58-59
        super().__init__(model, alphabet=alphabet, name=name, batch_size=batch_size, epochs=epochs, device=device)
