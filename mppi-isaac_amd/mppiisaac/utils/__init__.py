"""Config store, tensor transport, planner RPC and actor helpers (reference mppiisaac/utils)."""
