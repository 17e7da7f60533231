"""`compressai.models.base.CompressionModel` (`model/model.py:7,15,34,41`, `model/entropy_models.py:7,128`)."""
import torch.nn as nn

from ..entropy_models import EntropyBottleneck, GaussianConditional, get_scale_table


class CompressionModel(nn.Module):
    def aux_loss(self):
        """Sum of the quantile losses of every EntropyBottleneck (`train.py:230`)."""
        return sum(m.loss() for m in self.modules() if isinstance(m, EntropyBottleneck))

    def update(self, scale_table=None, force=False):
        """Install the Gaussian scale table and refresh the factorised-prior tables (`model/model.py:30-34`)."""
        if scale_table is None:
            scale_table = get_scale_table()
        updated = False
        for m in self.modules():
            if isinstance(m, EntropyBottleneck):
                updated |= m.update(force=force)
            elif isinstance(m, GaussianConditional):
                updated |= m.update_scale_table(scale_table, force=force)
        return updated
