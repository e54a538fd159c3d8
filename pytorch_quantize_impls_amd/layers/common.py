"""Layer plumbing shared by every quantised layer family."""
import torch


class QLayer():
    """Marker mixin used by model-surgery tooling (reference: QuantTorch/layers/common.py:1-9)."""

    def get_quant_weight(self):
        raise NotImplementedError

    def set_quant_weight(self):
        raise NotImplementedError

    def restore_weight(self):
        raise NotImplementedError


class EvalSwapMixin:
    """The reference's train()/eval() protocol, identical in every layer family
    (layers/binary_layers.py:30-40, terner_layers.py:30-40, dorefa_layers.py:29-39):

      * train -> eval : stash the real weight in the non-persistent attribute ``weight.org`` and
        overwrite ``weight.data`` with its quantised image;
      * eval -> train : copy ``weight.org`` back.

    ``state_dict()`` taken in eval mode therefore holds the QUANTISED weight (upstream hazard,
    SURVEY.md section 5) — kept bit-for-bit.  On top of it the mixin keeps a cache of the packed
    bit planes of the eval-mode weight, keyed on the weight's version counter, so eval-mode
    forwards of a device layer never re-pack.
    """

    def _quantized_weight_for_eval(self) -> torch.Tensor:  # pragma: no cover - abstract
        raise NotImplementedError

    def train(self, mode=True):
        if self.training == mode:
            return self
        self.training = mode
        self._qt_eval_planes = None
        if mode:
            self.weight.data.copy_(self.weight.org.data)
        else:
            if not hasattr(self.weight, 'org'):
                self.weight.org = self.weight.data.clone()
            self.weight.org.data.copy_(self.weight.data)
            with torch.no_grad():
                self.weight.data.copy_(self._quantized_weight_for_eval().detach())
            self._qt_eval_version = self.weight._version
        return self

    def _eval_cache(self) -> dict:
        """Cache of everything derived from the eval-mode weight, valid for one (version counter, storage) of it:
        load_state_dict(), optimizer steps, copy_() and .to(device) invalidate it.  Writes through ``weight.data``
        do NOT bump the version counter — after such an edit call ``reset_quant_cache()``."""
        w = self.weight
        cache = getattr(self, "_qt_eval_planes", None)
        if not isinstance(cache, dict) or cache.get("version") != w._version or cache.get("ptr") != w.data_ptr():
            cache = {"version": w._version, "ptr": w.data_ptr()}
            self._qt_eval_planes = cache
        return cache

    def reset_quant_cache(self):
        """Drop the packed planes / scales cached for the eval-mode weight (needed after a manual ``weight.data`` edit)."""
        self._qt_eval_planes = None

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self._qt_eval_planes = None

    def _eval_planes(self, packer, key="valu"):
        """Packed image of the eval-mode (already quantised) weight in the operand format ``key``;
        rebuilt if the weight tensor was written since eval() (load_state_dict, manual edits,
        .to(device))."""
        cache = self._eval_cache()
        if key not in cache:
            w = self.weight
            cache[key] = packer(w.detach().reshape(w.shape[0], -1))
        return cache[key]

    def _weight_on_grid(self, w: torch.Tensor) -> torch.Tensor:   # pragma: no cover - abstract
        """0-dim bool tensor: every entry of ``w`` is a value this family's quantiser can produce."""
        raise NotImplementedError

    def _eval_on_grid(self) -> bool:
        """True iff the eval-mode weight really holds a quantised image.  The reference's eval forward is just
        F.linear / F.conv2d on ``weight`` (layers/binary_layers.py:46), whatever it holds — e.g. a float checkpoint
        loaded AFTER .eval().  The packed device paths re-apply the quantiser, so they are only taken for an on-grid
        weight; an off-grid one goes through the dense expression like upstream.  One device reduction + one host sync
        per weight version (the cache above)."""
        cache = self._eval_cache()
        if "on_grid" not in cache:
            w = self.weight.detach()
            cache["on_grid"] = bool(self._weight_on_grid(w).item()) if w.is_cuda and w.numel() else True
        return cache["on_grid"]
