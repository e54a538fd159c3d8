"""Global default device, mirroring QuantTorch/device.py:2 ("use the GPU if there is one")."""
import torch

device = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
