"""Deployment-side helpers."""
