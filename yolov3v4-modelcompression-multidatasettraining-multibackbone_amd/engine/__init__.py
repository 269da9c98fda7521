"""HIP execution engine: ctypes binding (hiplib), graph lowering (plan) and batched NMS (nms)."""
