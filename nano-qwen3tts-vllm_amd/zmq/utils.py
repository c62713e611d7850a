"""Port probing for the worker sockets (reference: P/zmq/utils.py)."""
import socket


def find_available_port(start_port: int = 9600, max_attempts: int = 100, span: int = 4) -> int:
    """First port p >= start_port such that p .. p+span-1 are all free on 127.0.0.1."""
    for base in range(start_port, start_port + max_attempts * span, span):
        ok = True
        for p in range(base, base + span):
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
                try:
                    s.bind(("127.0.0.1", p))
                except OSError:
                    ok = False
                    break
        if ok:
            return base
    raise RuntimeError(f"no {span} consecutive free ports found from {start_port}")
