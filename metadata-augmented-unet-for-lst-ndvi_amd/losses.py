"""Mirror of the reference's MSE criterion (src/utils/losses.py:27-39) on the HIP path."""
from .functional import MSELoss


def compute_loss_mse(outputs, targets):
    """Same return contract as the reference: {'total': loss, 'mse': loss} (src/utils/losses.py:34-39)."""
    mse = MSELoss.apply(outputs, targets)
    return {"total": mse, "mse": mse}
