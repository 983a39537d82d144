python tools/jobs/job25.py
