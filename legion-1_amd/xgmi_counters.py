"""xGMI byte counters of the node, read from the driver's gpu_metrics table through amdsmi ("xGMI from
amd-smi / rocm-smi per-link byte counters sampled around the timed region").  Measurement plumbing for bench.py only:
best effort -- any failure (amdsmi missing, counters "N/A", no permission) yields None and the bench line says so."""
from __future__ import annotations

import sys

_state = {"amdsmi": None, "failed": False}


def _lib():
    if _state["failed"]:
        return None
    if _state["amdsmi"] is None:
        try:
            try:
                import amdsmi
            except ImportError:
                sys.path.append("/opt/rocm/share/amd_smi")
                import amdsmi
            amdsmi.amdsmi_init()
            _state["amdsmi"] = amdsmi
        except Exception:  # noqa: BLE001
            _state["failed"] = True
            return None
    return _state["amdsmi"]


def read():
    """{"read_kb": total, "write_kb": total, "gpus": n, "links": n_links_with_data} summed over every GPU and xGMI link of
    the node (xgmi_read_data_acc / xgmi_write_data_acc: accumulated kilobytes per link), or None."""
    smi = _lib()
    if smi is None:
        return None
    try:
        tot_r = tot_w = links = 0
        handles = smi.amdsmi_get_processor_handles()
        for h in handles:
            m = smi.amdsmi_get_gpu_metrics_info(h)
            r = [x for x in (m.get("xgmi_read_data_acc") or []) if isinstance(x, int)]
            w = [x for x in (m.get("xgmi_write_data_acc") or []) if isinstance(x, int)]
            tot_r += sum(r)
            tot_w += sum(w)
            links += sum(1 for x in r if x > 0)
        return {"read_kb": tot_r, "write_kb": tot_w, "gpus": len(handles), "links": links}
    except Exception:  # noqa: BLE001
        return None


def rate(before, after, seconds, n_gpus):
    """Average per-GPU xGMI read / write GB/s between two read() samples, or None."""
    if not before or not after or seconds <= 0 or n_gpus <= 0:
        return None
    return {"read_GBps_per_gpu": round((after["read_kb"] - before["read_kb"]) * 1024 / seconds / 1e9 / n_gpus, 2),
            "write_GBps_per_gpu": round((after["write_kb"] - before["write_kb"]) * 1024 / seconds / 1e9 / n_gpus, 2),
            "gpus_seen": after["gpus"], "seconds": round(seconds, 3),
            "source": "amdsmi gpu_metrics xgmi_read_data_acc / xgmi_write_data_acc, every GPU and link of the node, sampled by rank 0 around the timed windows"}
