@echo off
rem UTAU / OpenUtau launcher of the MI355X resampler backend (arguments pass through unchanged)
cd /d "%~dp0"
python SillySampler.py %*
