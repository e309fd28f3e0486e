#!/usr/bin/env python3
"""Drop-in for the reference's `fenics_mcp_server.py`: the orchestrator starts `python <cwd>/fenics_mcp_server.py`
as an MCP stdio server (multi_agent_orchestrator.py:27, 70-78) and the dispatcher calls tools by name
(dispatcher_agent.py:111-112, 210).  This file registers the pdegpu tools under the same names and runs the
same stdio transport (fenics_mcp_server.py:1899, 4554-4555).  Nothing is ever printed to stdout except by the
MCP transport itself."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))

from pdegpu import server as _server  # noqa: E402
from pdegpu.fields import PlotResult, SolveResult, TimeSeriesField  # noqa: E402,F401  (pickle class path of the serving process)

for _name, _fn in _server.TOOLS.items():
    globals()[_name] = _fn


def main() -> None:
    try:
        from mcp.server.fastmcp import FastMCP
    except ImportError as e:   # the `mcp` package is part of the reference's environment, not of this library
        raise SystemExit("fenics_mcp_server.py (pdegpu): the `mcp` package is required to serve tools over stdio "
                         f"({e}); the tool functions themselves are importable from pdegpu.server") from e
    mcp = FastMCP("FEniCS-Heat")          # same server name as fenics_mcp_server.py:1899
    for fn in _server.TOOLS.values():
        mcp.tool()(fn)
    mcp.run(transport="stdio")


if __name__ == "__main__":
    main()
