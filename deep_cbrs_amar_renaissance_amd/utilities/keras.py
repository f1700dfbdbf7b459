"""Parameter counting used by the driver — mirrors `/root/reference/src/utilities/keras.py:10-22`."""


def get_total_parameters(model):
    """(trainable, non-trainable) parameter counts of a model."""
    trainable = sum(int(p.numel()) for p in model.trainable_weights)
    non_trainable = sum(int(p.numel()) for p in model.non_trainable_weights)
    return trainable, non_trainable
