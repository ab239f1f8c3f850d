"""Deterministic synthetic inputs for the BASELINE.json configs (no datasets exist offline).

* json_tiles      -- configs[1]/[4]: benches/compression_66k_JSON.txt tiled cyclically.
* log_stream      -- configs[3]: the reference's hdfs.json / log corpus is absent from the mount, so the 8 GiB
                     "synthetic log stream" is defined HERE: fixed-width log lines whose fields come from a
                     counter-based generator (splitmix64 of the line index), so any byte range can be produced
                     independently on any rank / device.  md5 of the first 1 MiB is pinned in tests.
Everything is torch tensor code (runs on CPU for tests and on the GPU for the benchmark).
"""
import torch

LINE = 128   # bytes per log line, '\n' included

_LEVELS = [b"INFO ", b"DEBUG", b"WARN ", b"ERROR", b"TRACE"]
_LEVEL_CDF = [70, 85, 93, 98, 100]          # Zipf-ish
_SVCS = [b"auth", b"cart", b"feed", b"mail", b"pays", b"rank", b"srch", b"user"]
_WORDS = [b"accounts", b"articles", b"balances", b"checkout", b"comments", b"contacts", b"devices_", b"invoices",
          b"messages", b"networks", b"payments", b"products", b"profiles", b"projects", b"sessions", b"settings",
          b"shipment", b"tracking", b"vouchers", b"webhooks", b"wishlist", b"workflow", b"catalogs", b"channels",
          b"clusters", b"couriers", b"datasets", b"exports_", b"features", b"gateways", b"handlers", b"incident"]
_STATUS = [b"200", b"201", b"204", b"301", b"304", b"400", b"401", b"403", b"404", b"500", b"502", b"503"]
_STATUS_CDF = [70, 75, 78, 80, 84, 87, 89, 91, 96, 98, 99, 100]


def _i64(v):
    """wrap a python int to the int64 range"""
    return ((v + (1 << 63)) % (1 << 64)) - (1 << 63)


def _hash64(idx, stream):
    """counter-based 63-bit hash of (idx, stream): xorshift-multiply rounds in wrapping int64 arithmetic"""
    def k(v):
        return torch.tensor(_i64(v), dtype=torch.int64, device=idx.device)
    c1, c2, c3 = 0x9E3779B97F4A7C15, 0xBF58476D1CE4E5B9, 0x94D049BB133111EB
    x = idx * k(c1) + k((stream + 1) * c2)
    x = (x ^ ((x >> 30) & 0x3FFFFFFFF)) * k(c2)
    x = (x ^ ((x >> 27) & 0x1FFFFFFFFF)) * k(c3)
    x = x ^ ((x >> 31) & 0x1FFFFFFFF)
    return x & 0x7FFFFFFFFFFFFFFF


def _digits(val, width):
    """int64 tensor [n] -> uint8 [n, width] zero-padded decimal ASCII"""
    out = []
    for _ in range(width):
        out.append((val % 10 + 48).to(torch.uint8))
        val = val // 10
    return torch.stack(out[::-1], dim=1)


def _hexd(val, width):
    out = []
    for _ in range(width):
        d = val % 16
        out.append(torch.where(d < 10, d + 48, d + 87).to(torch.uint8))
        val = val // 16
    return torch.stack(out[::-1], dim=1)


def _table(rows, device):
    return torch.tensor([list(r) for r in rows], dtype=torch.uint8, device=device)


def _pick_cdf(r100, cdf, device):
    c = torch.tensor(cdf, dtype=torch.int64, device=device)
    return (r100.unsqueeze(1) >= c.unsqueeze(0)).sum(dim=1)


def log_lines(first_line, n_lines, device="cpu"):
    """uint8 tensor [n_lines * LINE]: lines first_line .. first_line + n_lines of the synthetic log stream"""
    dev = torch.device(device)
    idx = torch.arange(first_line, first_line + n_lines, dtype=torch.int64, device=dev)
    h = [_hash64(idx, s) for s in range(6)]
    ts = 1_700_000_000_000 + idx * 37 + (h[0] % 29)                       # epoch ms, monotonic
    level = _table(_LEVELS, dev)[_pick_cdf(h[1] % 100, _LEVEL_CDF, dev)]
    svc = _table(_SVCS, dev)[(h[1] // 100) % len(_SVCS)]
    pod = _digits((h[1] // 1000) % 24, 2)
    req = _hexd(h[2] % (1 << 24), 16)                                     # 24 bits of request id: repeats inside the window
    user = _digits(h[3] % 2_000, 10)                                      # a bounded user population: repeats
    ver = _digits(1 + (h[4] % 3), 1)
    word = _table(_WORDS, dev)[(h[4] // 3) % len(_WORDS)]
    rid = _digits((h[4] // 100) % 1000, 5)
    status = _table(_STATUS, dev)[_pick_cdf(h[5] % 100, _STATUS_CDF, dev)]
    dur = _digits((h[5] // 100) % 500, 5)

    def lit(b):
        return torch.tensor(list(b), dtype=torch.uint8, device=dev).unsqueeze(0).expand(n_lines, -1)

    parts = [_digits(ts, 13), lit(b" "), level, lit(b" ["), svc, lit(b"-"), pod, lit(b"] req="), req, lit(b" user="), user,
             lit(b" path=/api/v"), ver, lit(b"/"), word, lit(b"/"), rid, lit(b" status="), status, lit(b" dur="), dur,
             lit(b"ms")]
    line = torch.cat(parts, dim=1)
    pad = LINE - 1 - line.shape[1]
    assert pad >= 0, line.shape
    if pad:
        line = torch.cat([line, lit(b" " * pad)], dim=1)
    line = torch.cat([line, lit(b"\n")], dim=1)
    return line.reshape(-1)


def log_stream(byte_offset, n_bytes, device="cpu", chunk_lines=1 << 18):
    """bytes [byte_offset, byte_offset + n_bytes) of the log stream (any range, any device)"""
    assert byte_offset % LINE == 0 and n_bytes % LINE == 0
    first = byte_offset // LINE
    n = n_bytes // LINE
    out = torch.empty(n_bytes, dtype=torch.uint8, device=device)
    done = 0
    while done < n:
        m = min(chunk_lines, n - done)
        out[done * LINE:(done + m) * LINE] = log_lines(first + done, m, device)
        done += m
    return out


def json_tiles(plain, total, phase=0, device="cpu"):
    """buf[i] = plain[(i + phase) mod len(plain)], i < total  """
    jt = torch.frombuffer(bytearray(plain), dtype=torch.uint8).to(device)
    phase %= len(plain)
    reps = (total + phase) // len(plain) + 2
    return jt.repeat(reps)[phase:phase + total].contiguous()
