import torch
a = torch.load("gpurun_out/rz_row.pt"); b = torch.load("gpurun_out/rz_cell.pt")
for k in a:
    x, y = a[k], b[k]
    nan = int(torch.isnan(y).sum()); neq = int((x != y).sum() - (torch.isnan(x) & torch.isnan(y)).sum())
    print(k, "nan in cell out:", nan, "mismatching elements:", neq, "max abs diff:", float((x - y).nan_to_num().abs().max()))
