"""Small model-walking helpers (reference: QuantTorch/utils/tools.py:20-31)."""
import torch


def flat_net(model, class_to_get):
    """Every sub-module of ``model`` that is an instance of ``class_to_get``, in definition order;
    a matching module is returned whole (its children are not searched), like upstream."""
    if isinstance(model, class_to_get):
        return [model]
    found = []
    for child in model.children():
        if isinstance(child, class_to_get):
            found.append(child)
        elif isinstance(child, torch.nn.Module):
            found.extend(flat_net(child, class_to_get))
    return found
